#!/usr/bin/env python3
"""Benchmark of the segmenting watershed transform on MI355X (BASELINE.json metric: Mpixels/s, % of HBM roofline).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config headline|c4|c5]

headline (default)  one 8192x8192 u8 random field per GPU, device-resident: BASELINE's metric.  N GPUs = N independent
                    fields, no collective on the data path ("weak").
c3                  BASELINE config 3: the MERGING transform of one 8192x8192 field per GPU as the reference defines its
                    output: transform_to_list, the lake sizes of all 255 levels (lib.rs:1551-1561, tests/core_bench.rs:48),
                    records left in HBM.  c3-final: the final canonical labels only (`transform` is a stub in the reference).
c4                  BASELINE config 4: a batch of 64 independent 4096x4096 slices, slice i on rank i % N, each rank's
                    slices as ONE stacked transform (ws_segment_batch_device).  Total work fixed ("strong").
c5                  BASELINE config 5: one 32768x32768 field in row blocks over the ranks: ws_segment_tiled_device, the loop
                    inside the library (csrc/ws_tiled.hip), halo rows / flag / tables through RCCL when every rank has a
                    GPU of its own; `--local-ranks R` on one process: R virtual ranks on one device.  "strong".

A "step" is one pass of the whole hot path over the configuration's input (seed tables, all 255 water levels, final
labels); image and seeds are resident in HBM before the timed region, u32 labels stay in HBM.
headline and c3 keep `--contexts` (4) such steps in flight: engine contexts on streams of their own take turns
(ws_segment_device_begin / _end), so that a step is queued before the one ahead of it has been waited for and steps of
different contexts overlap on the GPU.  `value` is the throughput of that; the time of ONE transform with nothing else in
flight is measured right after and reported as config.ms_one_transform_alone (`--no-pipeline`: only that form).

Launch: under torchrun (RANK / WORLD_SIZE in the environment) every process is one rank.  A plain
`python bench.py --gpus N` with N > 1 starts its own N rank processes BEFORE touching a GPU.  When the ranks
outnumber the visible GPUs (a rehearsal on a one-GPU box) they share devices and the collectives run over gloo.

Rank 0 prints ONE JSON line (contract in the round brief) with `roofline`, `cpu_baseline`, and, for the headline
configuration on one GPU, `end_to_end` (the host ABI, PCIe included) and `secondary` (a smooth map)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
SWEEP_BYTES_PER_PX = 255 * 5 + 4   # per level one u8 + one u32 read, each label written once
C4_SLICES, C4_SIDE = 64, 4096
C5_SIDE = 32768


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=20)      # transforms are 0.6 ms: twenty bring the clocks up and let the graph capture (2nd call) settle
    ap.add_argument("--config", choices=["headline", "c3", "c3-final", "c4", "c5"], default="headline")
    ap.add_argument("--size", type=int, default=0, help="override the field side (headline: 8192, c4 slices: 4096, c5: 32768)")
    ap.add_argument("--slices", type=int, default=C4_SLICES, help="c4: slices in the batch")
    ap.add_argument("--engine", choices=["fused", "sweep"], default="fused")
    ap.add_argument("--cpu-size", type=int, default=4096, help="side of the CPU-baseline sample field (0 = skip)")
    ap.add_argument("--cpu-runs", type=int, default=3)
    ap.add_argument("--no-cpu-full", action="store_true", help="skip the one CPU run of the bench field itself (~45 s at 8192^2)")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end / secondary / cpu_baseline")
    ap.add_argument("--local-ranks", type=int, default=1, help="c5 on ONE process: virtual ranks of the local group (1: the field whole)")
    ap.add_argument("--tiles", default="", help="c5: PYxPX -- the field cut in both directions (ws_segment_tiled2d_device) instead of row blocks; PY * PX = the ranks")
    ap.add_argument("--contexts", type=int, default=4, help="headline: engine contexts that take turns, each on its own stream (1: as --no-pipeline)")
    ap.add_argument("--one-stream", action="store_true", help="headline: all contexts on one stream (transforms queue up, never overlap)")
    ap.add_argument("--no-pipeline", action="store_true", help="headline: one context, every transform waited for before the next is queued (the default takes turns on two contexts: ws_segment_device_begin / _end)")
    return ap.parse_args()


# ---- self-launch: `python bench.py --gpus N` without torchrun ---------------------------------------------------------

def spawn_ranks(n):
    """Starts n copies of this command line as ranks 0..n-1 (fresh interpreters: nothing here has touched a GPU) and
    waits for them; rank 0's stdout (the JSON line) passes through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


# ---- CPU baseline -----------------------------------------------------------------------------------------------------

def cpu_baseline(size, seed, runs, full_size=0):
    """The oracle's rayon-shaped port (oracle/ws_oracle_par.c: parallel full-image scan -> sequential scatter, per
    level, as lib.rs:1689-1748) on this host's cores; median of `runs` full transforms of a size x size sample, and
    (full_size > size) ONE transform of the bench field itself beside it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    def timed(side, n_runs):
        img = ol.random_field(side, side, seed)
        seeds = ol.find_local_minima(img)
        times, scans = [], 0
        for _ in range(max(n_runs, 1)):
            t0 = time.perf_counter()
            _, st = ol.segment_par(img, seeds)
            times.append(time.perf_counter() - t0)
            scans = st.scans
        times.sort()
        return times, scans
    times, scans = timed(size, runs)
    med = times[len(times) // 2]
    out = {"value": round(size * size / med / 1e6, 4), "unit": "Mpixels/s", "cores": ol.max_threads(), "kind": "port",
           "sample": f"{size}x{size} u8 field of the bench's generator (seed {seed}), full 255-level segmenting transform "
                     f"({scans} full-image scans), median of {len(times)} runs",
           "seconds_per_run": [round(t, 2) for t in times]}
    if full_size > size:
        ft, fscans = timed(full_size, 1)
        out["bench_field_run"] = {"value": round(full_size * full_size / ft[0] / 1e6, 4), "unit": "Mpixels/s", "seconds": round(ft[0], 2),
                                  "sample": f"the bench field itself: {full_size}x{full_size}, seed {seed}, one full transform ({fscans} scans)"}
    return out


# ---- extras of the headline configuration -----------------------------------------------------------------------------

def end_to_end(pkg, eng, H, W, runs=5):
    """ws_segment through the host ABI with reused, already touched host buffers: pageable u8 image and u64 seeds in,
    u64 labels out -- what a Rust caller of transform(ArrayView2<u8>, &seeds) (lib.rs:1810) sees.  Never `value`."""
    import ctypes
    import importlib
    import numpy as np
    ffi = importlib.import_module("rustronomy_watershed_amd._ffi")
    img = eng.random_field(H, W, 1).cpu().numpy()      # the engine's generator (ws_random_field_device) and one D2H copy
    ws = pkg.TransformBuilder.default().build_segmenting()
    ctx, opt = ws._ctx(), ws._opt
    cap = (H // 2 + 1) * (W // 2 + 1)
    seeds = np.zeros((cap, 2), dtype=np.uint64)
    labels = np.zeros((H, W), dtype=np.uint64)
    n = ctypes.c_size_t(0)
    rc = ffi.lib().ws_find_local_minima(ctx.handle, img.ctypes.data, H, W, W, seeds.ctypes.data, cap, ctypes.byref(n))
    assert rc == 0, rc
    times = []
    for i in range(runs + 2):
        t0 = time.perf_counter()
        rc = ffi.lib().ws_segment(ctx.handle, img.ctypes.data, H, W, W, seeds.ctypes.data, n.value, ctypes.byref(opt), labels.ctypes.data)
        dt = time.perf_counter() - t0
        assert rc == 0, rc
        if i >= 2:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    # bytes that cross the bus: the image, the seeds as they are and the labels as u32 (widened by host threads into the caller's
    # u64 plane: csrc/ws_hostcopy.hip)
    nbytes = H * W + n.value * 16 + H * W * 4
    nbytes_host = H * W + n.value * 16 + H * W * 8
    # the same with u32 labels out (ws_segment_u32): what a caller that can hold 4-byte labels pays
    labels32 = np.zeros((H, W), dtype=np.uint32)
    t32 = []
    for i in range(runs + 2):
        t0 = time.perf_counter()
        rc = ffi.lib().ws_segment_u32(ctx.handle, img.ctypes.data, H, W, W, seeds.ctypes.data, n.value, ctypes.byref(opt), labels32.ctypes.data)
        dt = time.perf_counter() - t0
        assert rc == 0, rc
        if i >= 2:
            t32.append(dt)
    t32.sort()
    med32 = t32[len(t32) // 2]
    same = bool((labels32 == labels.astype(np.uint32)).all())
    # ... and the README's call pair (lib.rs:73-86) as one call, ws_segment_minima: no seed list in either direction
    def timed(fn, out):
        ts = []
        for i in range(runs + 2):
            t0 = time.perf_counter()
            rc = fn(ctx.handle, img.ctypes.data, H, W, W, ctypes.byref(opt), out.ctypes.data, None, 0, ctypes.byref(n2))
            dt = time.perf_counter() - t0
            assert rc == 0, rc
            if i >= 2:
                ts.append(dt)
        ts.sort()
        return ts[len(ts) // 2]
    n2 = ctypes.c_size_t(0)
    lab_m = np.zeros((H, W), dtype=np.uint64)
    med_m = timed(ffi.lib().ws_segment_minima, lab_m)
    same_m = bool((lab_m == labels).all()) and n2.value == n.value
    med_m32 = timed(ffi.lib().ws_segment_minima_u32, labels32)
    t0 = time.perf_counter()
    rc = ffi.lib().ws_find_local_minima(ctx.handle, img.ctypes.data, H, W, W, seeds.ctypes.data, cap, ctypes.byref(n))
    ms_minima = (time.perf_counter() - t0) * 1e3
    assert rc == 0, rc
    ctx.close()
    pair = {"entry_point": "ws_segment_minima (find_local_minima + transform as one call: u8 image in, u64 labels out, no seed list)",
            "ms": round(med_m * 1e3, 3), "bytes_over_pcie": int(H * W * 5), "pcie_GBps": round(H * W * 5 / med_m / 1e9, 1),
            "bytes_in_the_callers_buffers": int(H * W * 9),
            "equal_to_the_two_calls": same_m, "ms_u32_labels": round(med_m32 * 1e3, 3),
            "the_two_calls_ms": round(ms_minima + med * 1e3, 3), "ws_find_local_minima_ms": round(ms_minima, 3)}
    return {"entry_point": "ws_segment (host ABI: pageable u8 image + u64 seed pairs in, u64 labels out)",
            "call_pair_as_one_call": pair,
            "ms": round(med * 1e3, 3), "Mpixels_per_s": round(H * W / med / 1e6, 1), "bytes_over_pcie": int(nbytes),
            "pcie_GBps": round(nbytes / med / 1e9, 1), "bytes_in_the_callers_buffers": int(nbytes_host),
            "runs": len(times), "host_buffers": "reused, touched",
            "u32_labels": {"entry_point": "ws_segment_u32 (the same, u32 labels out)", "ms": round(med32 * 1e3, 3),
                           "bytes_over_pcie": int(nbytes), "pcie_GBps": round(nbytes / med32 / 1e9, 1), "equal_to_u64_labels": same},
            "note": "the link is the bound (pageable copies run at the box's 52-54 GB/s, the transform is 0.56 ms of a call): u64 labels "
                    "cross it as u32 and are widened by four host threads while the next chunk is in flight "
                    "(ws_ctx_set_host_threads; 0 = one 8-byte copy of a plane widened on the device, as before: 11.5 ms for the one-call form)"}


def host_cube(pkg, cube, H, W, runs=3):
    """c4 from HOST memory: a cube of slices as the reference's own tests walk it (tests/integration.rs:267,356: find_local_minima
    + transform per slice) -- the loop of ws_segment_minima calls against ONE ws_segment_batch (the slices take turns on internal
    contexts: upload / transform / label copy of different slices overlap).  u8 slices in, u64 label planes out.  Never `value`."""
    import ctypes
    import importlib
    import numpy as np
    ffi = importlib.import_module("rustronomy_watershed_amd._ffi")
    L = ffi.lib()
    ws = pkg.TransformBuilder.default().build_segmenting()
    ctx, opt = ws._ctx(), ws._opt
    N = cube.shape[0]
    cube = np.ascontiguousarray(cube)
    a = np.zeros((N, H, W), dtype=np.uint64)
    b = np.zeros((N, H, W), dtype=np.uint64)
    n, failed = ctypes.c_size_t(0), ctypes.c_size_t(0)

    def loop():
        for k in range(N):
            assert L.ws_segment_minima(ctx.handle, cube[k].ctypes.data, H, W, W, ctypes.byref(opt), a[k].ctypes.data, None, 0, ctypes.byref(n)) == 0

    def batch():
        assert L.ws_segment_batch(ctx.handle, cube.ctypes.data, N, H, W, W, H * W, None, None, ctypes.byref(opt), b.ctypes.data, None, ctypes.byref(failed)) == 0

    def med(fn):
        ts = []
        for i in range(runs + 1):
            t0 = time.perf_counter()
            fn()
            if i >= 1:
                ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e3
    ms_loop, ms_batch = med(loop), med(batch)
    same = bool((a == b).all())
    ctx.close()
    return {"workload": f"{N} slices of {H}x{W} in host memory, seeds = each slice's find_local_minima, u64 label planes out",
            "ms_loop_of_ws_segment_minima": round(ms_loop, 2), "ms_ws_segment_batch": round(ms_batch, 2),
            "Mpixels_per_s": round(N * H * W / ms_batch / 1e3, 1), "same_labels": same,
            "note": "the link is the bound: 4 B/px of labels down and 1 B/px of image up; one call keeps both directions busy"}


def call_pair_device(eng, torch, H, W, runs=9):
    """The README's call pair on device-resident buffers (lib.rs:73-86): find_local_minima + transform as two calls, and as
    ONE (ws_segment_minima_device: the seed tables come out of the minima kernels, no list is written).  One context,
    nothing else in flight."""
    img = eng.random_field(H, W, 1)
    labels = torch.empty((H, W), dtype=torch.int32, device=eng.device)

    def med(fn):
        ts = []
        for i in range(runs + 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            if i >= 3:
                ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e3
    two = med(lambda: eng.segment(img, eng.find_local_minima(img), out=labels))
    ref = labels.clone()
    one = med(lambda: eng.segment_minima(img, out=labels))
    same = bool((labels == ref).all().item())
    one_list = med(lambda: eng.segment_minima(img, out=labels, want_seeds=True))
    return {"ms_two_calls": round(two, 4), "ms_one_call": round(one, 4), "ms_one_call_with_the_list": round(one_list, 4), "same_labels": same,
            "note": "two calls: ws_find_local_minima_device (writes the 56 MiB list, reads its count back) + ws_segment_device (k_seed_tables searches "
                    "and checks the list); one call: count, scan, compaction write the seed tables directly"}


def secondary_smooth(eng, torch, size, corr=64, runs=3):
    """The reference's real inputs are smooth maps (CGPS cube slices, Gaussian fields: tests/integration.rs:267-602), not
    iid noise: low-pass noise with a correlation length of `corr` pixels, generated on the device."""
    g = torch.Generator(device=eng.device).manual_seed(3)
    low = torch.rand((1, 1, size // corr + 2, size // corr + 2), device=eng.device, generator=g)
    up = torch.nn.functional.interpolate(low, size=(size, size), mode="bicubic", align_corners=False)[0, 0]
    up = (up - up.min()) / (up.max() - up.min())
    img = (up * 253.0).to(torch.uint8).contiguous()
    del low, up
    seeds = eng.find_local_minima(img)
    labels = torch.empty((size, size), dtype=torch.int32, device=eng.device)
    for _ in range(2):
        eng.segment(img, seeds, out=labels)
    torch.cuda.synchronize()
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        eng.segment(img, seeds, out=labels)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    st = eng.stats()
    return {"workload": f"{size}x{size} u8 smooth field (bicubic low-pass noise, correlation length {corr} px), "
                        f"{int(seeds.shape[0])} seeds = find_local_minima, segmenting, device-resident",
            "ms": round(med * 1e3, 3), "Mpixels_per_s": round(size * size / med / 1e6, 1),
            "relax_passes": int(st["relax_passes"]), "coloured_px": int((labels != 0).sum().item())}


# ---- one rank -----------------------------------------------------------------------------------------------------------

def run(args):
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (torchrun --nproc-per-node {args.gpus}), "
                         f"or run `python bench.py --gpus {args.gpus}` without WORLD_SIZE set and it starts its own ranks")
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev          # one rank per GPU on a real node; ranks share devices in a rehearsal
    backend = None
    comm_dev = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        # RCCL refuses two ranks on one device: a rehearsal with more ranks than GPUs runs its collectives over gloo
        backend = os.environ.get("WS_BENCH_BACKEND", "nccl" if torch.cuda.device_count() >= world else "gloo")
        dist.init_process_group(backend)
        comm_dev = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")

    if rank == 0:
        ge.build_hip()             # a no-op when the in-tree .so is current
    if world > 1:
        dist.barrier()
    pkg = ge.load_package()
    import importlib
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    wsd = importlib.import_module("rustronomy_watershed_amd.distributed")
    wsg = importlib.import_module("rustronomy_watershed_amd.group")
    torch.cuda.set_device(dev_index)
    # a real stream for everything (torch's default is the legacy null stream, on which nothing can be captured): the
    # engine replays the first passes of a transform that repeats the previous one's buffers as one hipGraph launch
    torch.cuda.set_stream(torch.cuda.Stream(dev_index))
    eng = dev.DeviceEngine(dev_index, engine=pkg.ENGINE_SWEEP if args.engine == "sweep" else pkg.ENGINE_FUSED)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the configuration's input, resident in HBM, and its step ------------------------------------------------------
    cfg = args.config
    pipelined = False
    c3_info = None
    to_list = cfg == "c3"            # BASELINE config 3 as the reference defines the merging transform's output (below)
    if cfg == "c3-final":
        cfg = "c3"
    if cfg in ("headline", "c3"):
        H = W = args.size or 8192
        img = eng.random_field(H, W, 1 + 16 * rank)     # independent fields per rank (and per context: below)
        seeds = eng.find_local_minima(img)
        labels = torch.empty((H, W), dtype=torch.int32, device=eng.device)
        n_seeds = int(seeds.shape[0])
        px_per_step_all_ranks = world * H * W
        scaling = "weak"
        merging = cfg == "c3"       # segmenting flood + one union pass over the image + relabel
        if to_list:
            # The merging transform's only real output in the reference is transform_to_list (lib.rs:1551-1561; what
            # tests/core_bench.rs:48 times): (level, lake sizes) for all 255 levels -- `transform` itself is a stub
            # (lib.rs:1524-1536).  One step = ws_transform_to_list_device: the flood, the per-level unions and the sparse
            # (colour, area) records of every level, left in HBM; only the 256 offsets and 255 uncoloured counts reach the host.
            lakes_buf, offsets, _unc = eng.transform_to_list(img, seeds, merging=True)
            n_records = int(offsets[-1])
            lakes_buf = torch.empty((n_records + 1024, 2), dtype=torch.int64, device=eng.device)
            seg = eng.segment(img, seeds)
            a, b = seg[:, :-1], seg[:, 1:]
            n_edges = int(((a != b) & (a != 0) & (b != 0)).sum().item())
            a, b = seg[:-1, :], seg[1:, :]
            n_edges += int(((a != b) & (a != 0) & (b != 0)).sum().item())
            del seg, a, b
            c3_info = {"records": n_records, "crossing_edges": n_edges, "levels": 255}

            def step():
                eng.transform_to_list(img, seeds, merging=True, lakes=lakes_buf)
        elif args.no_pipeline or args.contexts < 2 or args.engine == "sweep":
            def step():
                (eng.merge if merging else eng.segment)(img, seeds, out=labels)
        else:
            # `--contexts` engine contexts take turns (ws_segment_device_begin / _end), each on a stream of its own: transform
            # k is queued before transform k - 1 has been waited for, and transforms of different contexts may overlap on
            # the GPU -- where one transform leaves CUs idle (its latency-bound launches: seam strips, passes 2-4, the
            # chase) another one's kernels run.  Every step is still one whole transform of the same field into its
            # context's own label plane; a context's previous transform is waited for before its next one is queued.
            # (`--one-stream`: all contexts on one stream, so that transforms only queue up behind each other.)
            # EVERY context transforms a field and a seed list of its own (generator seeds 1 + 16 rank + i): the contexts'
            # inputs are distinct buffers with distinct contents (4 x 120 MiB > the 256 MiB Infinity Cache), so that no
            # transform re-reads what another one has just fetched.
            pipe_engines, pipe_labels, pipe_imgs, pipe_seeds = [eng], [labels], [img], [seeds]
            for i in range(1, args.contexts):
                if args.one_stream:
                    pipe_engines.append(dev.DeviceEngine(dev_index, engine=pkg.ENGINE_FUSED))
                else:
                    with torch.cuda.stream(torch.cuda.Stream(dev_index)):
                        pipe_engines.append(dev.DeviceEngine(dev_index, engine=pkg.ENGINE_FUSED))
                pipe_labels.append(torch.empty_like(labels))
                pipe_imgs.append(eng.random_field(H, W, 1 + 16 * rank + i))
                pipe_seeds.append(eng.find_local_minima(pipe_imgs[i]).clone())
            torch.cuda.synchronize()
            pipe = {"k": 0, "pending": [False] * args.contexts}
            for e_, l_, i_, s_ in zip(pipe_engines, pipe_labels, pipe_imgs, pipe_seeds):      # third call on: the context replays its graph
                for _ in range(3):
                    (e_.merge if merging else e_.segment)(i_, s_, out=l_)
            torch.cuda.synchronize()

            def step():
                i = pipe["k"] % args.contexts
                if pipe["pending"][i]:
                    (pipe_engines[i].merge_end if merging else pipe_engines[i].segment_end)()
                (pipe_engines[i].merge_begin if merging else pipe_engines[i].segment_begin)(pipe_imgs[i], pipe_seeds[i], pipe_labels[i])
                pipe["pending"][i] = True
                pipe["k"] += 1

            def drain():
                for i in range(args.contexts):
                    if pipe["pending"][i]:
                        (pipe_engines[i].merge_end if merging else pipe_engines[i].segment_end)()
                        pipe["pending"][i] = False
                pipe["k"] = 0
            pipelined = True
        workload = (f"{H}x{W} u8 uniform[0,254) random field per GPU, {('MERGING transform, transform_to_list: lake records of all 255 levels (ws_transform_to_list_device)' if to_list else 'MERGING transform (final canonical labels)') if cfg == 'c3' else 'segmenting transform'}, max_water_level 254, "
                    f"seeds = find_local_minima ({n_seeds} on rank 0's first field), engine {args.engine}"
                    + (f"; {args.contexts} contexts in flight, each transforming ITS OWN field and seed list (generator seeds 1..{args.contexts})" if pipelined else ""))
        parallelism = f"independent fields x{world}"
        units = {"fields_per_gpu": 1}
    elif cfg == "c4":
        H = W = args.size or C4_SIDE
        mine = wsd.shard_slices(args.slices, rank, world)            # slice i -> rank i % world
        cube = torch.empty((len(mine), H, W), dtype=torch.uint8, device=eng.device)
        per_slice, offs = [], [0]
        for j, k in enumerate(mine):
            cube[j] = eng.random_field(H, W, 1 + k)
            s = eng.find_local_minima(cube[j])
            per_slice.append(s)
            offs.append(offs[-1] + int(s.shape[0]))
        seeds = torch.cat(per_slice).contiguous() if per_slice else torch.empty((0, 2), dtype=torch.int32, device=eng.device)
        labels = torch.empty((len(mine), H, W), dtype=torch.int32, device=eng.device)
        n_seeds = int(seeds.shape[0])
        px_per_step_all_ranks = args.slices * H * W
        scaling = "strong"

        # the batch goes through ws_segment_batch_group (csrc/ws_tiled.hip): this process drives ONE rank of the job (a local
        # group of one rank on its own device) -- independent slices need no exchange step, so no RCCL communicator is made
        # (`--local-ranks R`: the rank's slices as R stacks on R virtual ranks of the one device -- R transforms in flight)
        vr = max(min(args.local_ranks, len(mine)), 1)
        grp = wsg.Group.local(vr, [dev_index] * vr)
        parts = []
        for r in range(vr):
            j0, j1 = len(mine) * r // vr, len(mine) * (r + 1) // vr
            parts.append((cube[j0:j1], seeds[offs[j0]:offs[j1]].contiguous(), [o - offs[j0] for o in offs[j0:j1 + 1]], labels[j0:j1]))

        def step():
            if mine:
                grp.segment_batch(H, W, parts)
        workload = (f"batch of {args.slices} independent {H}x{W} u8 random slices (CGPS-like cube), segmenting, slice i on rank "
                    f"i % {world}; a rank's {len(mine)} slices run as {'one stacked transform' if vr == 1 else f'{vr} stacked transforms in flight on {vr} virtual ranks of its device'} (ws_segment_batch_group -> ws_segment_batch_device)")
        parallelism = f"independent slices, {len(mine)} per GPU x{world}, no data-path collective"
        units = {"slices_total": args.slices, "slices_per_gpu": len(mine)}
    else:   # c5
        H = W = args.size or C5_SIDE
        rounds_seen = []
        vranks = max(args.local_ranks, 1)
        use_c_group = world == 1 or backend == "nccl"
        if use_c_group:
            # ws_segment_tiled_device (csrc/ws_tiled.hip): the tiled loop runs inside the library.  One rank per process and
            # GPU -> an RCCL group (halo rows by ncclSend / ncclRecv, the flag by ncclAllReduce, tables by ncclAllGather over
            # xGMI); a single process -> a LOCAL group of --local-ranks virtual ranks on its one device (1: the whole field).
            if world > 1:
                grp = wsg.Group.rccl_over_torch(dev_index, rank, world, comm_dev)
                grp.selftest()
            else:
                grp = wsg.Group.local(vranks, [dev_index] * vranks)
            full = eng.random_field(H, W, 5)                # every rank generates the field and keeps its rows (+ halo rows)
            all_seeds = eng.find_local_minima(full).clone() # global list; a rank's seeds are a contiguous range of it
            n_all = int(all_seeds.shape[0])
            tiles2d = tuple(int(v) for v in args.tiles.lower().split("x")) if args.tiles else None
            if tiles2d:
                # the field cut in both directions (py x px tiles, halo rows and columns; the general form's block steps)
                assert tiles2d[0] * tiles2d[1] == grp.world, f"--tiles {args.tiles}: {grp.world} ranks"
                blocks, spans, keep = grp.make_blocks2d(full, all_seeds, tiles2d[0], tiles2d[1])
                r0, r1 = spans[0][0][0], spans[0][0][1]

                def step():
                    rounds_seen.append(grp.segment_tiled2d_device(H, W, tiles2d[0], tiles2d[1], blocks, n_seeds_total=n_all))
            else:
                blocks, spans, keep = grp.make_blocks(H, lambda lo, hi, r: full[lo:hi].contiguous() if grp.world > 1 else full, all_seeds)
                r0, r1 = spans[0][0], spans[0][1]
                if grp.world > 1:
                    del full

                def step():
                    rounds_seen.append(grp.segment_tiled_device(H, W, n_all, blocks))
            del all_seeds
            torch.cuda.empty_cache()
            labels = None
            entry = f"ws_segment_tiled2d_device ({args.tiles} tiles, halo rows and columns)" if tiles2d else "ws_segment_tiled_device"
            how = (f"{entry}, RCCL group of {world} ranks (grouped ncclSend/ncclRecv halos, 1-word ncclAllReduce" + ("" if tiles2d else ", ncclAllGather of the boundary table") + ")"
                   if world > 1 else f"{entry}, local group of {vranks} rank(s) on one device")
        else:
            # rehearsal with more ranks than GPUs: RCCL refuses two ranks on one device, so the caller-driven form
            # (distributed.py: the same block steps, collectives over gloo) stands in
            r0, r1, lo, hi = wsd.row_block(H, rank, world)
            full = eng.random_field(H, W, 5)
            all_seeds = eng.find_local_minima(full)
            block_img = full[lo:hi].contiguous()
            n_all = int(all_seeds.shape[0])
            sl, colours = wsd.local_seeds(all_seeds, lo, hi)
            del full, all_seeds
            torch.cuda.empty_cache()
            block = wsd.HipBlockEngine(eng, block_img, sl, colours)

            def step():
                _, rounds = wsd.segment_tiled(block, rank, world)
                rounds_seen.append(rounds)
            labels = None
            how = f"distributed.py over {backend} (ranks share devices: a protocol rehearsal)"
        n_seeds = n_all
        px_per_step_all_ranks = H * W
        scaling = "strong"
        shape_txt = f"{args.tiles} tiles with a 1-pixel halo ring" if (use_c_group and args.tiles) else f"row blocks of {r1 - r0} rows per rank with 1-row halos"
        workload = (f"one {H}x{W} u8 random field, segmenting, {shape_txt}; {how}")
        parallelism = f"{'2-D' if (use_c_group and args.tiles) else 'row-blocked'} tiles x{world if world > 1 else vranks}"
        units = {"rows_per_gpu": r1 - r0}

    for _ in range(args.warmup):
        step()
    if pipelined:
        drain()
    # ---- timed region: exactly `steps` steps, no per-launch events (they cost ~12 %) ----
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if pipelined:
        drain()             # the wait for the last two transforms is inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    st_last = eng.stats()
    replayed = bool(st_last.get("graph_launches", 0))      # of the last timed transform
    ms_one_context = None
    if pipelined:
        # the same K steps with NOTHING else in flight: every transform is waited for before the next is queued (what a
        # caller of the one-call form gets).  The transforms rotate over the contexts' fields (each context replays its own
        # graph on its own buffers), so that a transform never finds its input in the Infinity Cache from the one before.
        def step():
            i = pipe["k"] % args.contexts
            (pipe_engines[i].merge if merging else pipe_engines[i].segment)(pipe_imgs[i], pipe_seeds[i], out=pipe_labels[i])
            pipe["k"] += 1
        for _ in range(2 * args.contexts):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        ms_one_context = (time.perf_counter() - t1) * 1e3 / args.steps

        def step():      # the kernel leg below: context 0 on its own field
            (eng.merge if merging else eng.segment)(img, seeds, out=labels)
    # ---- kernel leg: the same `steps` steps again with a HIP-event pair around every launch (recorded on the stream
    # the kernels run on) for the roofline object ----
    agg = {"ms_relax": 0.0, "ms_resolve": 0.0, "ms_sweep": 0.0, "ms_other": 0.0, "ms_total": 0.0, "launches_relax": 0,
           "launches_resolve": 0, "launches_sweep": 0, "tiles_run_relax": 0, "relax_tile_iterations": 0}
    if cfg != "c5":
        eng.ctx.set_profiling(True)
        barrier()
        for _ in range(args.steps):
            step()
            st = eng.stats()
            for k in agg:
                agg[k] += st[k]
        barrier()
        eng.ctx.set_profiling(False)

    t = torch.tensor([dt], dtype=torch.float64, device=comm_dev if world > 1 else eng.device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    if rank == 0:
        ms_step = dt_max / args.steps * 1e3
        value = px_per_step_all_ranks * args.steps / dt_max / 1e6
        npx_rank = px_per_step_all_ranks // world if cfg not in ("headline", "c3") else H * W      # pixels this rank's step covers
        if cfg == "c4":
            npx_rank = len(mine) * H * W
        n_seeds_rank = n_seeds if cfg != "c5" else n_seeds // world
        # compulsory HBM bytes of one step on this rank: image read once (1 B/px), every stamp and label written once
        # (4 + 4 B/px), seeds read once (8 B as u32 pairs on the device; SURVEY 8d counts the host's 16 B)
        b_min = npx_rank * 9 + 16 * n_seeds_rank
        if cfg == "c3":      # + the segmenting labels read once and the union-find parents of the seed colours written once
            b_min += npx_rank * 4 + 4 * n_seeds_rank
        b_sweep = npx_rank * SWEEP_BYTES_PER_PX + 16 * n_seeds_rank
        compulsory_GBps = b_min / (ms_step * 1e-3) / 1e9
        sweep_equiv = b_sweep / (ms_step * 1e-3) / 1e9
        roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        if cfg != "c5":
            if args.engine == "fused":
                kname, k_ms, k_launches = "k_relax", agg["ms_relax"], agg["launches_relax"]
                tile_px = 256 * 32
                # every tile that runs reads its image (1 B) and stamps (4 B) once per pixel -- except in pass 0, which has no
                # stamps to read: it derives them from the seed bit plane (1/8 B per pixel); every stamp is written once, by
                # pass 0 (later passes rewrite only what changed).  These are the bytes the kernel ASKS for, re-runs of a
                # tile included: they are the honest numerator for this kernel's own bandwidth, not a claim that the re-runs
                # are needed -- `frac_compulsory` below is the figure against the bytes the transform cannot avoid.
                rows_stack = npx_rank // W
                tiles_pass0 = args.steps * ((W + 255) // 256) * ((rows_stack + 31) // 32)
                k_bytes = max(agg["tiles_run_relax"] - tiles_pass0, 0) * tile_px * 5 + tiles_pass0 * tile_px * 1.125 + args.steps * npx_rank * 4
            else:
                kname, k_ms, k_launches = "k_flood_step", agg["ms_sweep"], agg["launches_sweep"]
                k_bytes = agg["launches_sweep"] * npx_rank * (1 + 4 + 4)
            k_avg_ms = k_ms / max(k_launches, 1)
            k_bytes_per_launch = k_bytes / max(k_launches, 1)
            achieved = k_bytes_per_launch / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
            # the same kernel against the bytes its PHASE cannot avoid (every pixel's image byte, seed bit and stamp once):
            # what is left of `frac` when the re-runs of tiles are not counted as useful bytes
            k_min_bytes = args.steps * npx_rank * (1 + 0.125 + 4) if args.engine == "fused" else k_bytes
            k_min_GBps = k_min_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            roof.update({
                "kernel": kname, "achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "kernel_minimum_bytes_per_step": int(k_min_bytes / args.steps), "frac_kernel_minimum": round(k_min_GBps / HBM_PEAK_GBS, 5),
                "avg_launch_ms": round(k_avg_ms, 5), "launches_per_step": round(k_launches / args.steps, 2),
                "timing": "HIP events around every launch, in a second leg of the same K steps (events off in the timed leg)",
                "tiles_run_per_step": round(agg["tiles_run_relax"] / args.steps, 1),
                "tile_sweep_iterations_per_step": round(agg["relax_tile_iterations"] / args.steps, 1),
                "algorithmic_bytes_per_launch": int(k_bytes_per_launch),
                "note": "achieved/frac: bytes THIS kernel asks for (5 B per pixel of every tile run in passes >= 1 -- 8192-pixel tiles; the 256x8 bands of the seam repair count a quarter -- "
                        "counted on the device, re-runs included; 1.125 B per pixel in pass 0; 4 B per pixel written once) over "
                        "its HIP-event time -- DESIGN.md section 5.  The whole-transform figure against unavoidable bytes is "
                        "frac_compulsory.",
                "device_ms_per_step": {k: round(agg[k] / args.steps, 4) for k in ("ms_total", "ms_relax", "ms_resolve", "ms_sweep", "ms_other")},
            })
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if cfg == "c3":
                roof["note"] += "  (c3: k_relax is the dominant kernel of the merging transform too; the union / relabel part adds ~0.35 ms)"
            if cfg == "headline" and args.engine == "fused" and H == 8192 and os.path.exists(tpath):
                tj = json.load(open(tpath))
                roof["traffic"] = int(tj["k_relax"]["bytes_per_launch"])
                roof["traffic_source"] = "SINGLE-CONTEXT profile (one transform at a time, not the timed mode with several in flight): " + tj["source"]
                if "transform_total_bytes" in tj:
                    roof["traffic_total"] = int(tj["transform_total_bytes"])       # all kernels of one transform, PMC
                    roof["traffic_total_over_compulsory"] = round(tj["transform_total_bytes"] / b_min, 3)
                    roof["frac_traffic_total"] = round(tj["transform_total_bytes"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        else:
            roof.update({"kernel": "whole tiled transform", "achieved": round(compulsory_GBps, 2),
                         "frac": round(compulsory_GBps / HBM_PEAK_GBS, 5), "traffic": None,
                         "note": "compulsory bytes of this rank's row block over the step time (no per-kernel leg in tiled mode)",
                         "exchange_rounds_per_step": round(sum(rounds_seen) / max(len(rounds_seen), 1), 2)})
        # the box's own achievable HBM rate (report against both): a 1 GiB device-to-device copy, read + write
        a = torch.empty(1 << 28, dtype=torch.int32, device=eng.device)
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(5):
            b.copy_(a)
        torch.cuda.synchronize()
        copy_GBps = 5 * 2 * a.numel() * 4 / (time.perf_counter() - tc) / 1e9
        del a, b
        roof["copy_GBps_measured"] = round(copy_GBps, 1)
        roof["frac_compulsory_of_copy"] = round(compulsory_GBps / copy_GBps, 5)
        # whole transform against the bytes it cannot avoid -- the honest roofline fraction
        roof["compulsory_bytes_per_step"] = int(b_min)
        roof["compulsory_GBps"] = round(compulsory_GBps, 1)
        roof["frac_compulsory"] = round(compulsory_GBps / HBM_PEAK_GBS, 5)
        # speed-equivalent only (NOT a roofline fraction): how fast a literal 255-sweep engine would have to move its
        # bytes to finish in the same time; the figure BASELINE.md's ">= 30 %" target is phrased in
        roof["sweep_model_speed_equivalent"] = {"bytes_per_step": int(b_sweep), "equivalent_GBps": round(sweep_equiv, 1),
                                                 "times_hbm_peak": round(sweep_equiv / HBM_PEAK_GBS, 3),
                                                 "note": "speed-equivalent of a sweep-per-level engine, not bandwidth"}
        # The other ceiling: vector issue.  The transform's kernels execute `lane_instr_per_px` vector instructions
        # per pixel (SQ_INSTS_VALU, profiles/valu.json: a single-context profile of this build's kernels) and a SIMD spends
        # `cycles_per_wave_instr` cycles on each (SQ_ACTIVE_INST_VALU; the microbenchmark has the same figure for the relaxation's
        # mix: v_min3 / v_med3 / v_min issue at half rate on gfx950).  peak = 1024 SIMDs x 64 lanes x 2.4 GHz / that.
        roof_valu = None
        vpath = os.path.join(ROOT, "profiles", "valu.json")
        if cfg == "headline" and args.engine == "fused" and H == 8192 and os.path.exists(vpath):
            vj = json.load(open(vpath))
            cpi = float(vj["cycles_per_wave_instr_measured"])
            peak_mix = 1024 * 64 * 2.4e9 / cpi
            lane_ops = float(vj["lane_instr_per_px"]) * npx_rank
            roof_valu = {"peak_lane_ops_per_s": round(peak_mix, 0), "peak_note": f"1024 SIMDs x 64 lanes x 2.4 GHz / {cpi:.2f} cycles per wave-instruction of THIS instruction mix "
                                                                              "(full-rate instructions alone: 3.2 cycles measured, 2 by the data sheet)",
                         "lane_ops_per_px": round(float(vj["lane_instr_per_px"]), 1), "lane_ops_per_step": int(lane_ops),
                         "achieved_lane_ops_per_s": round(lane_ops / (ms_step * 1e-3), 0), "frac": round(lane_ops / (ms_step * 1e-3) / peak_mix, 4),
                         "frac_of_datasheet_peak": round(lane_ops / (ms_step * 1e-3) / (1024 * 64 * 2.4e9 / 2.0), 4),
                         "floor_ms_per_step": round(lane_ops / peak_mix * 1e3, 4), "source": vj["source"], "microbenchmark": vj["microbenchmark"]}
            if ms_one_context is not None:
                roof_valu["frac_one_transform_alone"] = round(lane_ops / (ms_one_context * 1e-3) / peak_mix, 4)
            roof["tighter_bound"] = "valu" if roof_valu["frac"] > roof.get("frac_compulsory", 0.0) else "hbm"
            roof["tighter_bound_note"] = ("pass 0 of the relaxation and k_resolve_local keep a vector instruction active on every SIMD 88 % / 83 % of their cycles "
                                          "(profiles/r3_v0_issue_counters.json): the transform is bound by vector issue, not by HBM; `bound` stays \"hbm\" because "
                                          "BASELINE's metric is phrased against the HBM roofline")
        out = {
            "metric": ("Mpixels/s segmenting watershed, 8192x8192 u8, device-resident"
                       + (f" (throughput, {args.contexts} transforms of distinct fields in flight; one transform alone: value_one_transform_alone)" if pipelined else "")) if cfg == "headline"
                      else (("Mpixels/s merging watershed, transform_to_list (lake sizes of all 255 levels, lib.rs:1551-1561), 8192x8192 u8, BASELINE config c3, device-resident"
                             if to_list else "Mpixels/s merging watershed (final canonical labels only), 8192x8192 u8, BASELINE config c3, device-resident") if cfg == "c3"
                            else f"Mpixels/s segmenting watershed, BASELINE config {cfg}, device-resident"),
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u8 image / u32 stamps+labels (integer min/max/compare)", "data": "synthetic",
            "config": dict({"workload": workload, "name": cfg, "parallelism": parallelism,
                            "collective_backend": backend, "world_size": world, "devices_visible": torch.cuda.device_count(),
                            "launch": (("every step runs all of its kernels; the seed tables, the first 5 passes and the resolve "
                                        "are replayed as one hipGraph because the buffers repeat (stream launches: +1-2 %)")
                                       if replayed else "stream launches") +
                                      (f"; {args.contexts} contexts take turns (ws_segment_device_begin / _end), "
                                       + ("all on one stream: a transform is queued behind the one before it before the host waits"
                                          if args.one_stream else
                                          "each on its own stream: a transform is queued before the one before it has been waited for, and "
                                          "transforms of different contexts overlap on the GPU (throughput, not the latency of one transform)")
                                       + f" -- one context, every transform waited for before the next is queued: {ms_one_context:.4f} ms per step"
                                       if ms_one_context is not None else "")},
                           **units),
            "roofline": roof,
            "roofline_valu": roof_valu,
        }
        if c3_info is not None:
            # compulsory bytes of the list part: every crossing edge read once (8 B), every lake record
            # written once (16 B), on top of the flood's own
            b_list = 8 * c3_info["crossing_edges"] + 16 * c3_info["records"]
            c3_info.update({"records_per_s": round(c3_info["records"] / (ms_step * 1e-3), 0),
                            "list_bytes_compulsory": int(b_list), "frac_compulsory_with_list": round((b_min + b_list) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})
            t1 = time.perf_counter()
            for _ in range(args.steps):
                eng.merge(img, seeds, out=labels)
            torch.cuda.synchronize()
            c3_info["final_labels_only_ms"] = round((time.perf_counter() - t1) * 1e3 / args.steps, 4)
            out["c3"] = c3_info
        if ms_one_context is not None:      # the time of ONE transform when nothing else is in flight (the one-call form)
            out["config"]["contexts_in_flight"] = args.contexts
            out["config"]["ms_one_transform_alone"] = round(ms_one_context, 4)
            out["ms_one_transform_alone"] = round(ms_one_context, 4)
            out["value_one_transform_alone"] = round(npx_rank * world / (ms_one_context * 1e-3) / 1e6, 2)
            roof["frac_compulsory_one_transform_alone"] = round(b_min / (ms_one_context * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        if labels is not None:
            out["config"]["coloured_px"] = int((labels != 0).sum().item())
        extras = cfg == "headline" and world == 1 and not args.no_extras
        if extras and args.engine == "fused":
            del labels
            out["secondary"] = secondary_smooth(eng, torch, H)
            # the same at other correlation lengths (4 px: ~680 k seeds, short floods; 256 px: one seed, one flood over the plane)
            out["secondary"]["other_correlation_lengths"] = {
                str(c): {k: v for k, v in secondary_smooth(eng, torch, H, corr=c).items() if k in ("ms", "relax_passes")} for c in (4, 16, 256)}
            torch.cuda.empty_cache()
            out["call_pair"] = call_pair_device(eng, torch, H, W)
            torch.cuda.empty_cache()
            out["end_to_end"] = end_to_end(pkg, eng, H, W)
        if cfg == "c4" and world == 1 and not args.no_extras:
            out["host_cube"] = host_cube(pkg, cube[:min(len(mine), 16)].cpu().numpy(), H, W)
        out["cpu_baseline"] = cpu_baseline(args.cpu_size, 1, args.cpu_runs, 0 if args.no_cpu_full or cfg != "headline" else H) if (world == 1 and args.cpu_size > 0 and not args.no_extras) else None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))       # before anything here has touched a GPU
    run(args)


if __name__ == "__main__":
    main()
