#!/usr/bin/env python3
"""Benchmark of the qvz encode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full pass of the hot path over a synthetic QGEN-1 quality matrix that is already resident in HBM:
k-means (to convergence) -> conditional-PMF counters (-> RCCL all-reduce when sharded) -> Lloyd-Max codebooks ->
WELL draws -> quantize (the -u product + coder feed) -> arithmetic coding into `shards` independent
reference-format streams (default: 8192 streams per GPU coded on the GPU, one lane per stream; --coder host uses
host threads instead; both produce byte-identical streams). Nothing is cached between steps.

Workload (BASELINE.json): 100M reads x 150 columns per GPU, flags of configs[2] (-c 1 -d L -f 0.5) by default;
--config 3 selects configs[3] (-c 3 -d M -r 1.0), --config 1 the 10M x 150 case. Reads shard across ranks (weak
scaling: every rank holds --reads reads); the only exchanges are the small all-reduces above.

Prints ONE JSON line on rank 0: metric/value/unit, ms_per_step, bits_per_symbol, per-phase seconds, a `roofline`
object for the dominant GPU kernel (HIP events around each launch on the launch stream) and a `cpu_baseline`
object (reference qvz, or the CPU oracle when the reference binary is absent, on a bounded sample).
Beside the headline number the line says what the product of a step costs outside the timed region: `product` (bytes on
disk of the shard set: codebooks once + streams; D2H and file assembly seconds, measured once after the timed steps),
`ingest` (text -> HBM, a bounded sample through the C driver's staging path), `strict` (the byte-identical single-stream
.q on a bounded sample) and a PMF pass over a full 72-symbol alphabet.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    1: dict(reads=10_000_000, cols=150, clusters=1, dist="M", mode="r", ratio=1.0, seed=2,
            name="10M reads x 150 cols, -c 1 -d M -r 1.0"),
    2: dict(reads=100_000_000, cols=150, clusters=1, dist="L", mode="f", ratio=0.5, seed=3,
            name="100M reads x 150 cols, -c 1 -d L -f 0.5"),
    3: dict(reads=100_000_000, cols=150, clusters=3, dist="M", mode="r", ratio=1.0, seed=3,
            name="100M reads x 150 cols, -c 3 -d M -r 1.0"),
}

# algorithmic bytes per symbol of each kernel (DESIGN.md "Kernels")
ALG_BYTES_PER_SYMBOL = {
    "pmf_hist": 1.0,             # each symbol read once (+1 B/read of labels when k > 1, added below)
    "kmeans_step": 1.0,          # one pass per iteration: each symbol read once (+1 B/read of labels written)
    "quantize": 5.0,             # symbol in, draw in, (qv, qi, st) out: the -u product and the coder's two planes
    "well_draws": 3.0,           # word stream written once and read once, draw matrix written once
    "pack_feed": 4.0,            # qi + st in, 2-byte feed entry out (host coder only)
    "model_records": 10.0,       # qi + st in, one packed 8-byte record out
    "code_records": 8.0,         # record in (+ ~0.17 B/symbol of stream out); bound by the serial interval chain per stream
}
# kernels whose time is a dependent-instruction chain per stream, not a bandwidth figure
CHAIN_KERNELS = ("code_records",)


def measured_traffic(kernel, args, cfg, shards, world):
    """HBM bytes per step of `kernel` from the committed PMC passes of this round (profiles/r05_pmc_traffic_cfgN.json, else round 4's), if
    they were taken on this very workload; None otherwise. (The counters need rocprofv3 around the process: tools/
    collect_profiles.sh runs this very script under it; bench.py itself cannot read them.)"""
    try:
        path = next(p for p in (os.path.join(ROOT, "profiles", f"r{rr:02d}_pmc_traffic_cfg{args.config}.json") for rr in (5, 4)) if os.path.exists(p))
        with open(path) as f:
            t = json.load(f)
        w = t["workload"]
        same = (w["config"] == args.config and w["reads"] == cfg["reads"] and w["columns"] == cfg["cols"] and
                w["coder"] == args.coder and w["shards"] == shards and world == 1 and w.get("qv_plane_written", False) == (not args.no_u))
        return t["kernels"][kernel]["hbm_bytes"] if same and kernel in t["kernels"] else None
    except (OSError, KeyError, ValueError, StopIteration):
        return None


def cpu_baseline(cfg, cores_note):
    """Reference qvz (oracle/_ref/qvz_ref, single thread - it has no threaded mode) on three bounded QGEN-1 samples of
    the same workload (60k, 240k and 2M reads: ~25 s of one core in all); falls back to the CPU oracle ("port") when the
    reference binary did not travel. The full workload's rate is an EXTRAPOLATION: a least-squares line through the three
    samples (seconds = fixed + per_symbol * symbols; the fixed part is the N-independent codebook generation)."""
    import oracle_lib as O
    c = cfg["cols"]
    flags = ["-c", str(cfg["clusters"]), "-d", cfg["dist"], "-" + cfg["mode"], str(cfg["ratio"])]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "qvz_ref")
    samples = [60_000, 240_000, 2_000_000]
    times = []
    kind = "reference" if os.path.exists(ref_bin) else "port"
    with tempfile.TemporaryDirectory() as td:
        for n in list(samples):
            if times and times[0] > 30.0 and n == samples[-1] and len(samples) == 3:
                samples = samples[:2]          # the N-independent part alone takes a minute (three clusters' codebooks): two samples bound the run
                break
            rows = O.qgen1_rows(cfg["seed"], n, c)
            if kind == "reference":
                inp = os.path.join(td, f"s{n}.txt")
                open(inp, "wb").write(rows.tobytes())
                del rows
                t = time.perf_counter()
                subprocess.run([ref_bin] + flags + [inp, os.path.join(td, "s.q")], check=True, capture_output=True, timeout=600)
                times.append(time.perf_counter() - t)
                os.remove(inp)
            else:
                t = time.perf_counter()
                O.encode(rows, k=cfg["clusters"], mode=O.MODE_RATIO if cfg["mode"] == "f" else O.MODE_FIXED,
                         ratio=cfg["ratio"], dist=cfg["dist"], want_u=False)
                times.append(time.perf_counter() - t)
    x = np.array(samples, dtype=np.float64) * c
    slope, fixed = np.polyfit(x, np.array(times), 1)     # seconds per symbol (N-dependent phases), N-independent seconds
    fixed = max(float(fixed), 0.0)
    full = cfg["reads"] * c
    # Three clusters at -r 1.0: the N-independent codebook generation takes the reference ~68 s, the two samples that bound the run
    # differ by less than its run-to-run noise, and a line through them says nothing about the per-symbol cost (it has come out
    # negative): no slope, no extrapolation then - `value` (the larger sample's measured rate) stands alone.
    measurable = slope > 0.0 and slope * x[-1] >= 0.05 * times[-1]
    n3, t3 = samples[-1], times[-1]
    others = "; ".join(f"{n_} reads {t_:.2f} s" for n_, t_ in zip(samples[:-1], times[:-1]))
    return dict(value=round(n3 * c / t3, 1), unit="quality-symbols/s", cores=1, kind=kind,
                sample=f"QGEN-1 seed {cfg['seed']}, {n3} reads x {c} cols, flags {' '.join(flags)}, whole encode() incl. codebook "
                       f"generation, {t3:.2f} s (`value`); also {others}; "
                       f"the full workload's figure below is a {len(samples)}-point extrapolation, not a run",
                sample_seconds={str(n_): round(t_, 3) for n_, t_ in zip(samples, times)},
                seconds_per_symbol=float(slope) if measurable else None, fixed_seconds=round(fixed, 3) if measurable else round(float(np.mean(times)), 3),
                extrapolated_full_workload_value=round(full / (fixed + slope * full), 1) if measurable else None,
                extrapolation_note=("least-squares line through the samples; the reference codes ONE stream, the headline "
                                    "`value` is for 8192 streams whose bits/symbol are higher (see bits_per_symbol_fewer_streams): "
                                    "a ratio of the two compares different products") if measurable else
                                   ("not extrapolated: the N-independent codebook generation is all these samples measure (their times differ by "
                                    "less than its noise), the per-symbol cost cannot be read off them"),
                host=cores_note)


def cpu_quota_cores(cores):
    """the CPU time this process may use per second of wall (cgroup v2 cpu.max / v1 cfs quota), in cores; `cores` when unlimited"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline_multi_core(cfg, cores):
    """The reference has no threaded mode; what a user with T cores can do is cut the file and run T copies.
    T = min(64, cores, the cgroup's CPU quota) copies of oracle/_ref/qvz_ref (the oracle when the reference did not travel), each on
    its own 500k-read piece of the workload's QGEN-1 stream, all started together: symbols of all pieces / wall. Every copy pays
    the N-independent codebook generation for its piece, as T separate files would."""
    import torch
    import oracle_lib as O
    from qvz_amd import kernels as K
    c = cfg["cols"]
    flags = ["-c", str(cfg["clusters"]), "-d", cfg["dist"], "-" + cfg["mode"], str(cfg["ratio"])]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "qvz_ref")
    kind = "reference" if os.path.exists(ref_bin) else "port"
    t_procs = max(1, min(64, cores, cpu_quota_cores(cores)))
    m = 500_000
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    with tempfile.TemporaryDirectory(dir=base) as td:
        for i in range(t_procs):                                 # the pieces' text, made on the GPU (the CPU generator takes 23 ns per symbol)
            K.untranspose_text(K.qgen1(cfg["seed"], m, c, first=i * m), m).cpu().numpy().tofile(os.path.join(td, f"p{i}.txt"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if kind == "reference":
            procs = [subprocess.Popen([ref_bin] + flags + [os.path.join(td, f"p{i}.txt"), os.path.join(td, f"p{i}.q")], stdout=subprocess.DEVNULL,
                                      stderr=subprocess.DEVNULL) for i in range(t_procs)]
        else:
            code = ("import sys, numpy as np; sys.path.insert(0, %r); import oracle_lib as O; rows = np.fromfile(sys.argv[1], dtype=np.uint8).reshape(-1, %d); "
                    "O.encode(rows, k=%d, mode=%d, ratio=%r, dist=%r, want_u=False)") % (os.path.join(ROOT, "tests"), c + 1, cfg["clusters"],
                                                                                      O.MODE_RATIO if cfg["mode"] == "f" else O.MODE_FIXED, cfg["ratio"], cfg["dist"])
            procs = [subprocess.Popen([sys.executable, "-c", code, os.path.join(td, f"p{i}.txt")]) for i in range(t_procs)]
        codes = [p.wait(timeout=900) for p in procs]
        wall = time.perf_counter() - t0
    if any(codes):
        return dict(failed=f"{sum(1 for x in codes if x)} of {t_procs} copies failed")
    return dict(value=round(t_procs * m * c / wall, 1), unit="quality-symbols/s", cores=t_procs, kind=kind, wall_seconds=round(wall, 2),
                sample=f"{t_procs} processes side by side, each the whole encode() of its own piece of {m} reads x {c} cols (QGEN-1 seed {cfg['seed']}, reads "
                       f"i * {m} ...), flags {' '.join(flags)}; {cores} host cores visible, CPU quota {cpu_quota_cores(cores)}",
                note="T separate files: T codebook sets, T streams - not the product the single-thread figure stands for; the honest many-core figure of a "
                     "program that has no threads")


def c_step(resident, sym, n, shards, cfg, mode, want_qv, first=0, n_total=None, group=None):
    """One step through the C driver (qvz_resident_encode, include/qvz_encode.h) -> the dict main() works with: phase seconds
    as the driver measured them, its own HIP-event kernel times (qvz_resident_products), the products' handle."""
    k = cfg["clusters"]
    r, p = resident.encode(sym, n, shards, clusters=k, mode=mode, ratio=cfg["ratio"], dist=cfg["dist"], want_qv=want_qv,
                           first_read=first, n_total=n_total, group=group)
    stats_in_cluster = r.t_stats if k == 1 else 0.0          # one cluster: the counters are made inside the clustering stage (they give it its sums)
    wall = dict(kmeans=r.t_cluster - stats_in_cluster, pmf=r.t_stats, codebook=r.t_codebook, quantize=r.t_quantize, coder=r.t_coder,
                coder_model_kernels=p.model_kernels_ms * 1e-3, coder_interval_kernels=p.coder_kernels_ms * 1e-3, total=r.t_total)
    kernels = dict(pmf_hist=[p.pmf_kernel_ms], quantize=[p.quantize_kernel_ms])
    if k > 1:
        kernels["kmeans_step"] = [p.kmeans_kernels_ms / max(1, r.kmeans_iterations)] * r.kmeans_iterations
    return dict(stream_bytes=None, total_stream_bytes=int(r.stream_bytes), distortion=r.distortion, kmeans_iterations=r.kmeans_iterations,
                deferred=r.deferred_decisions, wall=wall, kernels=kernels, products=p, resident=resident)


def product_report(r, n, c, k, shards, steps, elapsed):
    """What the streams of the last timed step cost as a product: packed on the device, copied to the host, written as ONE
    shard set (include/qvz_shardset.h: codebooks once + stream index + streams); measured once, outside the timed steps."""
    import torch
    from qvz_amd import host
    if "products" in r:          # the C driver's step: packed by the handle on request (qvz_resident_pack)
        p = r["products"]
        sizes = [int(p.h_sizes[g]) for g in range(p.shards)]
        handles = [p.codebooks[j] for j in range(p.clusters)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        packed = r["resident"].pack()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    else:
        sizes = r["stream_bytes"]
        handles = [cb.handle for cb in r["codebooks"]]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        packed = host.gather_streams(r["coded"], r["cap_per_shard"], sizes)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    pinned = torch.empty(packed.numel(), dtype=torch.uint8, pin_memory=True)
    t2 = time.perf_counter()
    pinned.copy_(packed, non_blocking=True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "out.qvzset")
        si = host.shardset_write(path, c, n, np.full(32, 0x55555555, dtype=np.uint32), handles, pinned, sizes)
        t4 = time.perf_counter()
        assert os.path.getsize(path) == si.file_bytes
    cb_bytes = int(si.codebook_bytes)
    ref_files = shards * (9 + cb_bytes + 128) + int(si.stream_bytes)     # the same streams as `shards` reference-format files
    sym = float(n) * c
    step_s = elapsed / steps
    return dict(container="one shard set (include/qvz_shardset.h): 160-byte header + 8 B per stream + the codebooks once + the streams",
                streams=shards, file_bytes_total=int(si.file_bytes), file_bits_per_symbol=round(si.file_bytes * 8.0 / sym, 5),
                stream_bytes=int(si.stream_bytes), codebook_bytes=cb_bytes, header_bytes=int(si.header_bytes),
                as_reference_format_shard_files_bytes=ref_files,
                as_reference_format_shard_files_bits_per_symbol=round(ref_files * 8.0 / sym, 5),
                pack_on_device_seconds=round(t1 - t0, 4), d2h_seconds=round(t3 - t2, 4), pinned_alloc_seconds=round(t2 - t1, 4),
                file_write_seconds=round(t4 - t3, 4), e2e_seconds=round(step_s + (t1 - t0) + (t3 - t2) + (t4 - t3), 4),
                e2e_note="ms_per_step + device pack + D2H + shard-set write to a temporary file (page cache); input already in HBM")


def ingest_report(cfg, n, c):
    """Text -> HBM through the C driver's staging path (qvz_ingest_rows: threaded host copy into pinned buffers, PCIe,
    transposing kernel, overlapped), on a bounded sample; the full workload's input is n * (c + 1) bytes."""
    import torch
    from qvz_amd import host
    from qvz_amd import kernels as K
    m = min(n, 4_000_000)
    sample = K.untranspose_text(K.qgen1(cfg["seed"], m, c), m).cpu()            # m rows of text in (pageable) host memory
    sym = K.empty_matrix(m, c)
    host.ingest_rows(sample, m, c, sym)                                          # first call: allocations, page faults
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bad = host.ingest_rows(sample, m, c, sym)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = m * (c + 1)
    full = n * (c + 1)
    return dict(sample_bytes=nbytes, seconds=round(dt, 4), gb_per_s=round(nbytes / dt / 1e9, 2), bad_bytes=bad,
                ingest_seconds=round(full / (nbytes / dt), 3), full_input_bytes=full,
                note="PCIe-inclusive; not part of `value` (inputs are resident in HBM when the timed region starts)")


def strict_report(cfg, c):
    """The byte-identical single-stream .q (the reference's own format: ONE serial arithmetic stream) through the C driver
    (qvz_encode_file) on a bounded sample file: the adaptive model runs on the GPU (a wave per column), one host core keeps
    the interval chain. Reported beside `value`, never part of it. `host_model` is the same with QVZ_STRICT_MODEL=host
    (model and chain both on the host core: the first round's strict path)."""
    import torch
    from qvz_amd import host
    from qvz_amd import kernels as K
    m = 2_000_000
    mode = host.MODE_RATIO if cfg["mode"] == "f" else host.MODE_FIXED
    out = {}
    with tempfile.TemporaryDirectory() as td:
        inp, q = os.path.join(td, "s.txt"), os.path.join(td, "s.q")
        K.untranspose_text(K.qgen1(cfg["seed"], m, c), m).cpu().numpy().tofile(inp)
        torch.cuda.synchronize()
        md5 = {}
        for label, env in (("gpu_model", None), ("host_model", "host")):
            if env:
                os.environ["QVZ_STRICT_MODEL"] = env
            else:
                os.environ.pop("QVZ_STRICT_MODEL", None)
            # several clusters: the initial centroids are libc rand() draws (cluster.c:198-201, never seeded: a fresh process - the
            # reference, the qvz binary - starts from srand(1)'s state); in THIS process whoever called rand() before has moved it
            # on, and the second file would start from other centroids than the first
            ctypes.CDLL(None).srand(1)
            t0 = time.perf_counter()
            r = host.encode_file(inp, q, clusters=cfg["clusters"], mode=mode, ratio=cfg["ratio"], dist=cfg["dist"], seed_mode=host.SEED_DEBUG)
            dt = time.perf_counter() - t0
            import hashlib
            md5[label] = hashlib.md5(open(q, "rb").read()).hexdigest()
            out[label] = dict(encode_phase_seconds=round(r.t_encode, 3), ns_per_symbol=round(r.t_encode / (m * c) * 1e9, 2),
                              whole_file_seconds=round(dt, 3), symbols_per_second=round(m * c / dt, 1), bits_per_symbol=round(r.stream_bytes * 8.0 / (m * c), 5))
        os.environ.pop("QVZ_STRICT_MODEL", None)
    out.update(sample=f"QGEN-1 seed {cfg['seed']}, {m} reads x {c} cols as a text file, whole qvz_encode_file incl. ingest and codebooks",
               identical_files=md5["gpu_model"] == md5["host_model"], streams=1,
               note="one reference-format .q with a single arithmetic stream; ns_per_symbol is the start_qv_compression phase "
                    "(GPU quantize + model, host interval chain, overlapped); compare cpu_baseline.seconds_per_symbol for the reference")
    return out


def other_configs_report(args, done_cfg, sym_done, threads):
    """The other single-GPU BASELINE configurations, a short run each (1 warm-up + 2 timed steps, same step as the headline:
    nothing cached between steps), so that one default invocation shows all of them. configs[3] re-uses the resident
    matrix when the headline ran configs[2] (same reads, other flags)."""
    import torch
    from qvz_amd import host
    from qvz_amd import kernels as K
    from qvz_amd import pipeline as P
    out = []
    for cid in sorted(CONFIGS, reverse=True):                    # the one that shares the resident matrix first
        if cid == args.config or args.reads:
            continue
        cfg = CONFIGS[cid]
        n, c, k = cfg["reads"] - cfg["reads"] % 4, cfg["cols"], cfg["clusters"]
        same = cfg["reads"] == done_cfg["reads"] and cfg["seed"] == done_cfg["seed"] and c == done_cfg["cols"]
        sym = sym_done if same else K.qgen1(cfg["seed"], n, c)
        mode = host.MODE_RATIO if cfg["mode"] == "f" else host.MODE_FIXED

        resident = host.Resident(torch.cuda.current_device()) if args.driver == "c" else None

        def step():
            if resident is not None:
                return c_step(resident, sym, n, 8192, cfg, mode, not args.no_u)
            return P.encode_pass(sym, n, 0, n, None, clusters=k, mode=mode, ratio=cfg["ratio"], dist=cfg["dist"], shards=8192,
                                 threads=threads, coder="gpu", keep_device=not args.no_u)

        torch.cuda.empty_cache()                                 # (what the extras before this one left in torch's cache has the wrong block sizes)
        r = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        each = []
        for _ in range(2):
            r = None                                             # the previous step's device outputs (qv plane, coded streams) make room first
            r = step()
            each.append(round(r["wall"]["total"] * 1e3, 1))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        rec = dict(config=cid, workload=cfg["name"], ms_per_step=round(dt * 1e3, 3), step_ms=each, value=round(n * c / dt, 1),
                   bits_per_symbol=round(r["total_stream_bytes"] * 8.0 / (n * c), 5), kmeans_iterations=r["kmeans_iterations"],
                   phase_seconds={ph: round(v, 4) for ph, v in r["wall"].items()})
        if "kernels" in r:                                       # the PMF pass of this configuration against the roofline (1 B per symbol, + 1 B per read of labels)
            pmf_ms = float(np.mean(r["kernels"]["pmf_hist"]))
            ach = (n * c + (n if k > 1 else 0)) / (pmf_ms * 1e-3) / 1e9
            rec["roofline_pmf_hist"] = dict(kernel="pmf_hist", bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                                            frac=round(ach / HBM_PEAK_GBPS, 4), ms_per_launch=round(pmf_ms, 4))
        out.append(rec)
        del r
        if resident is not None:
            resident.close()
        if not same:
            del sym
        torch.cuda.empty_cache()
    return out


def e2e_cli_report(cfg, sym, n, c, shards):
    """File to file through the C boundary, the way a user of the reference runs it: the resident matrix is written as the text
    file load_file would map (tmpfs), then `qvz -s -u out.u <flags> in.txt out.qset` (QVZ_SHARDS streams, one shard set) and
    `qvz -x out.qset out.dec` run as child processes - host C driver (qvz_amd/csrc/host/qvz_encode.c), ingest, -u text,
    shard-set write and process start-up (HIP runtime, module load) all inside the wall time; out.dec must equal out.u
    (the reference's test.sh:7-9). Reported beside `value`, never part of it."""
    import re
    import shutil
    import torch
    from qvz_amd import host
    from qvz_amd._lib import lib
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    need = 3 * n * (c + 1) + (n * c) // 2
    if shutil.disk_usage(base).free < need:
        return dict(skipped=f"needs {need >> 30} GiB under {base}")
    qvz_bin = os.path.join(ROOT, "qvz_amd", "bin", "qvz")
    flags = ["-c", str(cfg["clusters"]), "-d", cfg["dist"], "-" + cfg["mode"], str(cfg["ratio"])]
    td = tempfile.mkdtemp(prefix="qvz_e2e_", dir=base)
    try:
        inp, qset, u, dec = (os.path.join(td, x) for x in ("in.txt", "out.qset", "out.u", "out.dec"))
        t0 = time.perf_counter()
        host.egress_rows(sym, n, inp)
        t_in = time.perf_counter() - t0
        host.release_workspace()                                # the child process needs the HBM this one keeps between steps
        lib().qvz_hip_encode_planes_release()
        torch.cuda.empty_cache()
        env = dict(os.environ, QVZ_WELL_SEED="debug", QVZ_SHARDS=str(shards), QVZ_TIMING="1")
        settle = 6.0        # see `settle_note` below

        def timed(cmd, **kw):
            time.sleep(settle)
            t = time.perf_counter()
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, **kw)
            return res, time.perf_counter() - t

        def phases_of(res):
            last = [ln for ln in res.stderr.strip().splitlines() if ln.startswith("qvz_timing: upload=")][-1]
            return {k_: float(v) for k_, v in re.findall(r"(\w+)=([\d.]+(?:e[-+]?\d+)?)", last)}

        # with -u, without it twice (what a user who wants the .q only waits for), with -u again: the first child after this process
        # has given its HBM back still meets the allocation stall now and then, whatever it is asked to do; the faster run of each counts
        runs_u, runs = [], []
        for with_u in (True, False, False, True):
            cmd = [qvz_bin, "-s"] + (["-u", u] if with_u else []) + flags + [inp, qset if with_u else qset + ".nou"]
            r2, t2 = timed(cmd, env=env)
            if r2.returncode:
                return dict(failed=(r2.stdout + r2.stderr)[-600:])
            (runs_u if with_u else runs).append((t2, phases_of(r2)))
        t_enc, phases_u = min(runs_u, key=lambda x: x[0])
        t_enc_no_u, phases = min(runs, key=lambda x: x[0])
        same_set = subprocess.run(["cmp", "-s", qset, qset + ".nou"]).returncode == 0
        os.remove(qset + ".nou")
        dec_res, t_dec = timed([qvz_bin, "-x", qset, dec])
        if dec_res.returncode:
            return dict(failed=(dec_res.stdout + dec_res.stderr)[-600:])
        same = subprocess.run(["cmp", "-s", dec, u]).returncode == 0
        sym_count = float(n) * c
        device = sum(phases.get(x, 0.0) for x in ("cluster", "codebook", "quantize", "coder")) + (phases.get("stats", 0.0) if cfg["clusters"] > 1 else 0.0)
        keys = ("upload", "cluster", "stats", "codebook", "quantize", "coder", "egress", "encode", "total")
        return dict(command="QVZ_WELL_SEED=debug QVZ_SHARDS=%d qvz -s [-u out.u] %s in.txt out.qset" % (shards, " ".join(flags)),
                    input_bytes=os.path.getsize(inp), shard_set_bytes=os.path.getsize(qset), u_bytes=os.path.getsize(u),
                    encode_wall_seconds=round(t_enc_no_u, 3), encode_symbols_per_second=round(sym_count / t_enc_no_u, 1),
                    encode_wall_seconds_both_runs=[round(t, 3) for t, _ in runs],
                    encode_with_u_wall_seconds=round(t_enc, 3), encode_with_u_symbols_per_second=round(sym_count / t_enc, 1),
                    encode_with_u_wall_seconds_both_runs=[round(t, 3) for t, _ in runs_u],
                    same_shard_set_with_and_without_u=same_set,
                    phase_seconds={x: phases.get(x) for x in keys}, phase_seconds_with_u={x: phases_u.get(x) for x in keys},
                    device_part_seconds=round(device, 4),
                    device_part_note="k-means + PMF (inside `cluster` at one cluster) + codebooks + quantizer + coder kernels as the C driver schedules "
                                     "them in a one-shot process: compare ms_per_step (the same phases driven by qvz_amd/pipeline.py in a resident one)",
                    decode_wall_seconds=round(t_dec, 3), decode_symbols_per_second=round(sym_count / t_dec, 1),
                    decoded_equals_u=same, bits_per_symbol_on_disk=round(os.path.getsize(qset) * 8.0 / sym_count, 5),
                    input_text_write_seconds=round(t_in, 3), settle_seconds=settle,
                    settle_note="every child process starts %.0f s after the previous one has exited: a process that starts right after another "
                                "has freed ~180 GB of HBM stalls 1.5-3.5 s inside its first large hipMalloc (measured: profiles/r03_e2e_cli.txt), "
                                "which says nothing about this code" % settle,
                    note="tmpfs files (page cache; MAKING new tmpfs pages tops out at ~4-7 GB/s on this host whatever the thread count: the pages of "
                         "the -u text (15.1 GB) and of the shard set are made by a helper thread from the moment their size is known, the "
                         "writers copy through a mapping - host/qvz_egress.c, QVZ_PREALLOC=0 turns it off); child processes, HIP start-up included in the wall times; host threads: 8 copy "
                         "threads each way (QVZ_COPY_THREADS), one per cluster for codebooks, one for the file-order error sum")
    finally:
        shutil.rmtree(td, ignore_errors=True)


def pmf72_report(n, c):
    """The PMF pass over a matrix that uses all 72 symbols of the alphabet uniformly (QGEN-1 populates 40): the counter
    tiles are 72 x 72 instead of the populated sub-block."""
    import torch
    from qvz_amd import kernels as K
    sym = K.empty_matrix(n, c)
    step = 1 << 28
    flat = sym.view(-1)
    for a in range(0, flat.numel(), step):                                       # torch.randint in pieces (its int64 temporaries are large)
        flat[a:a + step] = torch.randint(0, 72, (min(step, flat.numel() - a),), dtype=torch.uint8, device=sym.device)
    cnt = torch.zeros((1, c, 72, 72), dtype=torch.int32, device=sym.device)
    K.pmf_hist(sym, n, cnt=cnt)
    ms = []
    for _ in range(3):
        cnt.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        K.pmf_hist(sym, n, cnt=cnt)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    assert int(cnt.sum(dtype=torch.int64).item()) == n * c
    ach = n * c / (min(ms) * 1e-3) / 1e9
    return dict(kernel="pmf_hist", alphabet=72, bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                frac=round(ach / HBM_PEAK_GBPS, 4), ms_per_launch=round(min(ms), 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--reads", type=int, default=0, help="override reads per GPU (testing)")
    ap.add_argument("--shards", type=int, default=0, help="arithmetic-coded streams per GPU (default: host threads)")
    ap.add_argument("--threads", type=int, default=0, help="host coder threads per rank (default: cores / ranks)")
    ap.add_argument("--coder", choices=["host", "gpu"], default="gpu",
                    help="arithmetic coder: host threads, or one GPU lane per shard (byte-identical streams)")
    ap.add_argument("--driver", choices=["c", "python"], default="c",
                    help="who runs the stages of the timed step: the C driver (qvz_resident_encode; GPU coder only) or qvz_amd/pipeline.py")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the product / ingest / strict / 72-symbol PMF measurements after the timed steps")
    ap.add_argument("--no-u", action="store_true", help="do not write the quantized-value plane (the -u product) in the timed step")
    ap.add_argument("--roundtrip", action="store_true",
                    help="after the timed steps, decode every stream on the GPU and compare with the quantizer's output")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from qvz_amd import host
    from qvz_amd import kernels as K
    from qvz_amd import pipeline as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # Rehearsal aid for a box with ONE GPU (the N > 1 launch line cannot be tried there otherwise): QVZ_BENCH_ONE_DEVICE=1 puts
    # every rank on cuda:0 and QVZ_BENCH_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one device). Never set
    # by the driver: on a multi-GPU node it is one rank per GPU over RCCL ("nccl").
    if os.environ.get("QVZ_BENCH_ONE_DEVICE"):
        local = 0
    backend = os.environ.get("QVZ_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    group = None
    if world > 1 or os.environ.get("QVZ_BENCH_FORCE_DIST"):   # (the variable: a one-rank RCCL group, to rehearse the N > 1 code path on one GPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    cfg = dict(CONFIGS[args.config])
    if args.reads:
        cfg["reads"] = args.reads
    n = cfg["reads"] - cfg["reads"] % 4            # per-rank reads; shard starts stay on WELL word boundaries
    c, k = cfg["cols"], cfg["clusters"]
    n_total = n * world
    first = rank * n
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # a GPU box's CPU share is 16 cores per GPU even where more logical CPUs are visible
    threads = args.threads or max(1, min(16, cores // max(1, world)))
    shards = args.shards or (threads if args.coder == "host" else 8192)
    mode = host.MODE_RATIO if cfg["mode"] == "f" else host.MODE_FIXED

    sym = K.qgen1(cfg["seed"], n, c, first=first)          # synthetic input, resident in HBM before timing
    torch.cuda.synchronize()
    means0 = None      # the k centroid reads the reference's rand() picks are fetched from the ranks that own them, inside the step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed step is the C driver's (qvz_resident_encode, include/qvz_encode.h: host code in C, HIP kernels behind the C-ABI);
    # torch supplies the resident matrix and - for N > 1 - the RCCL all-reduce the C side calls back into. --driver python runs
    # the same stages from qvz_amd/pipeline.py (what the tests drive with other backends and coders).
    use_c = args.driver == "c" and args.coder == "gpu"
    resident = host.Resident(local) if use_c else None

    def step_c():
        return c_step(resident, sym, n, shards, cfg, mode, not args.no_u, first=first, n_total=n_total, group=group)

    def step(timer=None):
        if use_c:
            return step_c()
        return P.encode_pass(sym, n, first, n_total, means0, clusters=k, mode=mode, ratio=cfg["ratio"], dist=cfg["dist"],
                             shards=shards, threads=threads, group=group, timer=timer, device_index=local,
                             coder=args.coder, keep_device=not args.no_u and args.coder != "host")

    for _ in range(args.warmup):
        step()
    timer = P.Timer(True)
    # Python's cyclic garbage collector stays out of the timed region: a full collection (deterministically around the 33rd
    # encode of a process, 30 - 50 ms with torch and numpy loaded) landed in whatever phase was allocating Python objects
    import gc
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    results = []
    for _ in range(args.steps):
        if results:                        # the device outputs of a finished step (qv plane, coded streams) make room for the next
            for key in ("coded", "qv", "labels"):
                results[-1].pop(key, None)
        results.append(step(timer))
        if os.environ.get("QVZ_BENCH_MEM"):                  # development: HBM head-room after every step
            free_b, total_b = torch.cuda.mem_get_info()
            print(f"step {len(results)}: torch allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB, reserved "
                  f"{torch.cuda.memory_reserved() / 2**30:.1f} GiB (peak {torch.cuda.max_memory_reserved() / 2**30:.1f}), device free "
                  f"{free_b / 2**30:.1f} of {total_b / 2**30:.1f} GiB", file=sys.stderr, flush=True)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        sb = torch.tensor([results[-1]["total_stream_bytes"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(sb)
        stream_bytes = int(sb.item())
    else:
        stream_bytes = results[-1]["total_stream_bytes"]

    roundtrip = None
    if args.roundtrip and args.coder != "host":
        # full-size parity property: encode -> decode of every stream reproduces the quantized values bit for bit
        draws = K.well_draws(np.full(32, 0x55555555, dtype=np.uint32), first, n, c)
        if use_c:                    # one more step of the C driver, keeping the quantized values; its products stay in the handle
            rr = c_step(resident, sym, n, shards, cfg, mode, True, first, n_total, group)
            p = rr["products"]
            ldq = sym.shape[1]
            coded = host.device_view(p.d_coded, p.cap_per_stream * shards)
            sizes = [int(p.h_sizes[g]) for g in range(shards)]
            want_qv = host.device_view(p.d_qv, c * ldq).view(c, ldq)
            want_lab = host.device_view(p.d_labels, n) if k > 1 else None
            handles, cap = [p.codebooks[j] for j in range(k)], int(p.cap_per_stream)
        else:
            r = P.encode_pass(sym, n, first, n_total, means0, clusters=k, mode=mode, ratio=cfg["ratio"], dist=cfg["dist"],
                              shards=shards, threads=threads, group=group, device_index=local, coder=args.coder, keep_device=True)
            coded, cap, sizes, want_qv, want_lab = r["coded"], r["cap_per_shard"], r["stream_bytes"], r["qv"], r["labels"]
            handles = [cb.handle for cb in r["codebooks"]]
        torch.cuda.synchronize()
        td = time.perf_counter()
        qv, lab = host.decode_shards_gpu(coded, cap, sizes, n, c, handles, shards, draws)
        torch.cuda.synchronize()
        td = time.perf_counter() - td
        same = bool(torch.equal(qv[:, :n], want_qv[:, :n])) and (k == 1 or bool(torch.equal(lab[:n], want_lab[:n])))
        roundtrip = dict(decoded_equals_quantized=same, decode_seconds=round(td, 4),
                         decode_symbols_per_second=round(n * c / td, 1), streams=shards)
        del qv, lab, draws
    if rank == 0:
        spans = timer.totals_ms()
        if use_c:                    # the C driver's own HIP events around its launches, on their stream (qvz_resident_products)
            for r in results:
                for name, v in r["kernels"].items():
                    spans.setdefault(name, []).extend(v)
        if args.coder == "gpu":      # the planes coder times its two kernel families itself (HIP events inside the C call)
            spans.pop("arith_shards", None)
            spans["model_records"] = [r["wall"]["coder_model_kernels"] * 1e3 for r in results]
            spans["code_records"] = [r["wall"]["coder_interval_kernels"] * 1e3 for r in results]
            if "coder_quantize_kernels" in results[-1]["wall"]:   # the quantizer as the coder pipeline's first stage: its launches, summed per step
                spans["quantize"] = [r["wall"]["coder_quantize_kernels"] * 1e3 for r in results]
        per_launch = {name: float(np.mean(v)) for name, v in spans.items()}
        per_step = {name: float(np.sum(v)) / args.steps for name, v in spans.items()}
        # `roofline` is reported for the dominant HBM-bound kernel. The lane-per-stream coder usually takes the most
        # time of all, but it is a serial integer chain per stream (latency / instruction-issue bound, DESIGN.md
        # section 6): an HBM fraction says nothing about it, so it is listed separately as `longest_kernel`.
        longest = max(per_step, key=per_step.get)
        streaming = {kname: v for kname, v in per_step.items() if kname not in CHAIN_KERNELS}
        dom = max(streaming, key=streaming.get)
        bps = ALG_BYTES_PER_SYMBOL[dom]
        if dom == "quantize" and (args.no_u or args.coder == "host"):
            bps = 4.0
        extra = 0.0
        if k > 1 and dom == "pmf_hist":
            extra = float(n)
        if k > 1 and dom == "kmeans_step":
            extra = float(n)
        alg_bytes = bps * n * c + extra
        achieved = alg_bytes / (per_launch[dom] * 1e-3) / 1e9
        roof = dict(kernel=dom, bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=measured_traffic(dom, args, cfg, shards, world),
                    ms_per_launch=round(per_launch[dom], 4), algorithmic_bytes_per_launch=int(alg_bytes),
                    bytes_per_symbol=bps, traffic_source=f"profiles/r05_pmc_traffic_cfg{args.config}.json, else r04 (rocprofv3 --pmc FETCH_SIZE / "
                    "WRITE_SIZE passes of this workload, FETCH_SIZE x2 per the gfx950 correction); null if the workload differs")
        pmf_ach = (n * c + (n if k > 1 else 0)) / (per_launch["pmf_hist"] * 1e-3) / 1e9
        wall = {ph: round(float(np.mean([r["wall"][ph] for r in results])), 4) for ph in results[-1]["wall"]}
        line = {
            "metric": "quality-symbols/s (encode)", "value": round(n_total * c * args.steps / elapsed, 1),
            "unit": "quality-symbols/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "step_ms": [round(r["wall"]["total"] * 1e3, 1) for r in results],
            "step_phase_ms": {ph: [round(r["wall"][ph] * 1e3, 1) for r in results] for ph in results[-1]["wall"] if ph != "total"},
            "step_kernel_ms": {name: [round(x, 1) for x in v] for name, v in spans.items() if len(v) == args.steps},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic QGEN-1 (SURVEY.md Appendix D), generated on device",
            "config": {"workload": cfg["name"] + f" per GPU, reads sharded over {world} GPU(s)", "reads_per_gpu": n,
                       "columns": c, "clusters": k, "coder": f"{shards} independent reference-format streams per GPU "
                       + (f"on {threads} host threads" if args.coder == "host" else
                          "coded on the GPU (model: a wave per stream and column; interval chain: a lane per stream)"),
                       "parallelism": f"read-sharded x{world}",
                       "host_driver": "C: qvz_resident_encode (include/qvz_encode.h)" if use_c else "Python: qvz_amd/pipeline.py encode_pass"},
            "bits_per_symbol": round(stream_bytes * 8.0 / (n_total * c), 5),
            "distortion": results[-1]["distortion"], "kmeans_iterations": results[-1]["kmeans_iterations"],
            "phase_seconds": wall,
            "gpu_kernel_ms_per_step": {kname: round(v, 4) for kname, v in sorted(per_step.items())},
            "roofline": roof,
            "roofline_pmf_hist": dict(kernel="pmf_hist", bound="hbm", achieved=round(pmf_ach, 1), peak=HBM_PEAK_GBPS,
                                      unit="GB/s", frac=round(pmf_ach / HBM_PEAK_GBPS, 4),
                                      ms_per_launch=round(per_launch["pmf_hist"], 4)),
            "longest_kernel": dict(kernel=longest, ms_per_step=round(per_step[longest], 3),
                                   bound="hbm" if longest not in CHAIN_KERNELS else
                                   "latency: one serial integer interval chain per stream (not an HBM or MFMA kernel)",
                                   symbols_per_second=round(n * c / (per_step[longest] * 1e-3), 1),
                                   streams=shards if longest in CHAIN_KERNELS else None),
            "coder_chain": dict(kernel="code_records" if args.coder == "gpu" else "arith_shards", streams=shards,
                                ns_per_symbol_per_stream=round(per_step.get("code_records", per_step.get("arith_shards", 0.0))
                                                               * 1e6 * shards / (n * c), 2)) if args.coder != "host" else None,
        }
        if args.coder == "gpu":
            # the coder phase against what SURVEY section 8(d) prices it at: 2 B per symbol in (quantizer index, state) + the streams out;
            # everything beyond that is the builder's own intermediate record (8 B per symbol written by the model kernel, read by the chain; 16 in rounds 1-3)
            tm, tc = measured_traffic("model_records", args, cfg, shards, world), measured_traffic("code_records", args, cfg, shards, world)
            survey = 2.0 * n * c + results[-1]["total_stream_bytes"]
            line["coder_phase_traffic"] = dict(hbm_bytes=(tm + tc) if tm and tc else None, survey_8d_bytes=int(survey),
                                               ratio=round((tm + tc) / survey, 2) if tm and tc else None,
                                               phase_ms=round(wall["coder"] * 1e3, 2),
                                               survey_bytes_over_phase_GBps=round(survey / wall["coder"] / 1e9, 1),
                                               note="model + chain kernels' HBM bytes (PMC, profiles/) over SURVEY 8(d)'s bytes for the phase: the records' "
                                                    "round trip through HBM is the price of splitting the adaptive model from the serial chain (DESIGN.md section 4)")
        if args.roundtrip and args.coder != "host":
            line["roundtrip"] = roundtrip
        line["qv_plane_written_in_step"] = not args.no_u and args.coder != "host"
        def guarded(name, fn, *a):
            # what comes after the timed steps must not be able to take the headline with it: an exception (or a child process
            # that does not come back) becomes an entry that says so
            try:
                line[name] = fn(*a)
            except Exception as exc:       # noqa: BLE001 - reported, not swallowed
                line[name] = dict(failed=f"{type(exc).__name__}: {exc}"[:600])

        if not args.no_extras and world == 1 and args.coder == "gpu" and ("coded" in results[-1] or "products" in results[-1]):
            guarded("product", product_report, results[-1], n, c, k, shards, args.steps, elapsed)
            for key in ("coded", "qv", "labels", "products", "resident"):
                results[-1].pop(key, None)
            if resident is not None:
                resident.close()                                  # its buffers (150+ GB) make room for what follows
            guarded("ingest", ingest_report, cfg, n, c)
            guarded("roofline_pmf_hist_72_symbols", pmf72_report, n, c)
            guarded("strict", strict_report, cfg, c)
            guarded("other_configs", other_configs_report, args, cfg, sym, threads)
            guarded("e2e_cli", e2e_cli_report, cfg, sym, n, c, shards)
        if not args.no_cpu_baseline and world == 1:               # the CPU baseline is timed on rank 0 of the 1-GPU run only
            guarded("cpu_baseline", cpu_baseline, CONFIGS[args.config], f"{cores} host cores visible, 1 used")
            if isinstance(line.get("cpu_baseline"), dict) and "failed" not in line["cpu_baseline"]:
                try:
                    line["cpu_baseline"]["multi_core"] = cpu_baseline_multi_core(CONFIGS[args.config], cores)
                except Exception as exc:   # noqa: BLE001 - reported, not swallowed
                    line["cpu_baseline"]["multi_core"] = dict(failed=f"{type(exc).__name__}: {exc}"[:400])
            if "gpu_model" in line.get("strict", {}) and line.get("cpu_baseline", {}).get("seconds_per_symbol"):   # the byte-identical single-stream file against the reference ON THIS HOST
                ref_ns = line["cpu_baseline"]["seconds_per_symbol"] * 1e9
                gpu_ns = line["strict"]["gpu_model"]["ns_per_symbol"]
                line["strict"]["reference_ns_per_symbol_same_host"] = round(ref_ns, 2)
                line["strict"]["speedup_vs_same_host_reference"] = round(ref_ns / gpu_ns, 2) if gpu_ns else None
                line["strict"]["speedup_note"] = ("per-symbol slope of the reference's encode (all N-dependent phases) over the strict path's "
                                                  "start_qv_compression phase, both on this host's cores; the format's one serial stream caps it")
        if "gpu_model" in line.get("strict", {}):
            # what sharding costs in bits: the same quantized symbols as ONE stream (the strict sample, 2M reads) and as the
            # sharded product's share of streams for that many reads would be unfair to compare; the full-size figures with fewer
            # streams are measurements of earlier rounds' sweeps (profiles/r01_a_bench_cfg2.jsonl, r02_stream_sweep_cfg2.jsonl)
            line["bits_per_symbol_single_stream_sample"] = line["strict"]["gpu_model"]["bits_per_symbol"]
        if args.config == 2 and not args.reads:
            line["bits_per_symbol_fewer_streams"] = {"16": 1.31253, "1024": 1.3231, "2048": 1.3312, "4096": 1.3449, "8192": 1.36743, "16384": 1.4036,
                                                     "source": "full-size runs of this workload: profiles/r01_a_bench_cfg2.jsonl (16 host-coded "
                                                               "streams), profiles/r02_stream_sweep_cfg2.jsonl; every stream re-learns the adaptive model"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
