#!/usr/bin/env python3
"""bench.py — headline benchmark of the antialiased resample hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json metric / configs[1]): uint8 channels_last [B,3,438,906] -> [196,320] bilinear antialias,
Pillow-exact arithmetic, B images per GPU resident in HBM before the timed region (B=1024 -> 1.22 GB in, far beyond
the 256 MB Infinity Cache).  A "step" is one pass of the hot path over one such batch.  Images shard across ranks
(weak scaling: fixed per-GPU batch); the only collective is the broadcast of the packed weight tables, outside
the timed region.  Prints ONE JSON line on rank 0.

roofline: dominant kernel's ALGORITHMIC bytes per launch (read input once + write output once = 1,378,644 B/image
for this config) ÷ its average duration measured here with HIP events on the launch stream.
cpu_baseline: the reference's own step_three separable C++ (oracle/_ref/ref_s3sep.so, built from /root/reference in
the build container) timed on this box's host cores on a bounded sample; falls back to our C port (kind "port").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H_IN, W_IN, H_OUT, W_OUT, CH = 438, 906, 196, 320, 3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, which also quotes a 6.29 TB/s float4-copy ceiling)
GUIDE_COPY_CEILING_GBS = 6290.0
PMC_SUMMARIES = {  # dominant kernel -> committed rocprofv3 --pmc summary (separate FETCH_SIZE / WRITE_SIZE passes), newest first
    "fused_u8_nhwc_pil_v3": ("r03_pmc_fused_v3.json", "r02_pmc_fused_v3.json", "r01_pmc_fused_v3.json"),
    "fused_u8_nhwc_pil": ("r01_pmc_fused_v1.json",),
}


def pmc_traffic_bytes(variant: str, batch: int):
    """(HBM bytes per launch of the dominant kernel, source file) from the committed PMC summary (rocprofv3 --pmc, separate
    passes for FETCH_SIZE and WRITE_SIZE, this bench's command at the summary's batch size).  Units: KiB.  gfx950 correction
    from MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 128-byte requests of a 16-B-per-lane stream as 64 B -> x2; WRITE_SIZE is
    exact.  The counters cannot be read from inside a process (they need rocprofv3 around it), so this is NOT live: the source
    file is named in the bench line, and traffic scales with the batch (every image is fetched and written exactly once).
    Third value: True when the summary is STALE — it carries the fingerprint of the kernel's sources at profiling time
    (tools/pmc_digest.py stamps it) and the sources have changed since, or it carries none; False when they match.
    (None, None, None) when no summary matches the kernel that ran."""
    from interpolate_antialiasing_amd import _lib

    now = _lib.source_fingerprint(variant)
    for name in PMC_SUMMARIES.get(variant, ()):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                d = json.load(f)
            fetch = d["FETCH_SIZE"]["avg_per_dispatch"]
            write = d["WRITE_SIZE"]["avg_per_dispatch"]
            pmc_batch = int(d.get("batch", 1024))
        except (OSError, KeyError, ValueError):
            continue
        stamped = d.get("source_fingerprint")
        stale = stamped is None or now is None or stamped != now
        return int((2.0 * fetch + write) * 1024 * batch / pmc_batch), f"profiles/{name} (batch {pmc_batch}, sources {stamped})", stale
    return None, None, None


def measure_copy_ceiling(dev, nbytes=2 << 30, reps=10):
    """Attainable HBM ceiling on THIS box: a device copy of `nbytes` (read + write = 2 x nbytes of traffic, far
    beyond the 256 MB Infinity Cache) by the library's 16-byte-per-lane copy kernels (aa_probe_copy, four forms), HIP events
    on the launch stream.  Returns (best GB/s of read + written bytes, {form: GB/s})."""
    from interpolate_antialiasing_amd import _lib

    L = _lib.load()
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 256)
    dst = torch.empty_like(src)
    stream = torch.cuda.current_stream(dev).cuda_stream
    forms = {}
    for form in range(6):
        for _ in range(3):
            _lib.check(L.aa_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, form, stream), "aa_probe_copy")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.aa_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, form, stream)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if form >= 4:  # one-way streams: bytes written (4) or read (5) per second
            forms["write_only" if form == 4 else "read_only"] = round(nbytes / (ms * 1e-3) / 1e9, 1)
            continue
        ok = bool(torch.equal(src[:1 << 20], dst[:1 << 20]) and torch.equal(src[-(1 << 20):], dst[-(1 << 20):]))
        if ok:
            forms[form] = round(2.0 * nbytes / (ms * 1e-3) / 1e9, 1)
        dst.zero_()
    del src, dst
    torch.cuda.empty_cache()
    copies = [v for k, v in forms.items() if isinstance(k, int)]
    return (max(copies) if copies else None), forms


def cpu_baseline(seconds: float = 12.0):
    """Reference CPU path on the host cores, bounded sample. Returns the cpu_baseline object."""
    import oracle

    # the GPU box exposes every host core (os.cpu_count() = 256) but gives a one-GPU job a 16-core share; more
    # OpenMP threads than that only oversubscribe (measured: 256 threads -> 0.5 s per image)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (1, CH, H_IN, W_IN), dtype=np.uint8)
    x8 = torch.from_numpy(img)
    mpix = H_IN * W_IN / 1e6
    ref = None
    try:
        ref = oracle.load_ref("ref_s3sep")
    except Exception:
        ref = None
    if ref is not None:
        kind = "reference"

        def run_u8():  # test.py:52-58,75: uint8 -> float -> op -> byte
            return ref.forward(x8.float(), [H_OUT, W_OUT], False).byte()

        xf = x8.float()

        def run_f32():
            return ref.forward(xf, [H_OUT, W_OUT], False)
    else:
        kind = "port"
        xf_np = img.astype(np.float32)

        def run_u8():
            return oracle.harness_u8("linear", img, (H_OUT, W_OUT), nthreads=cores)

        def run_f32():
            return oracle.forward("linear", xf_np, (H_OUT, W_OUT), nthreads=cores)

    def timeit(fn, budget):
        fn()
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget:
                return dt / n, n

    t_u8, n_u8 = timeit(run_u8, seconds * 0.4)
    t_f32, n_f32 = timeit(run_f32, seconds * 0.4)
    t_one = None
    if ref is not None:  # "at 1 thread and at all host cores"
        torch.set_num_threads(1)
        t_one, _ = timeit(run_u8, seconds * 0.2)
        torch.set_num_threads(cores)
    return {
        "value": round(mpix / t_u8, 2), "unit": "Mpix/s", "cores": cores, "kind": kind,
        "sample": (f"step_three -DUSE_SEPARABLE_KERNEL forward, [1,3,438,906]->[196,320] bilinear AA, uint8 via "
                   f"float()/byte() as test.py does, {n_u8} calls in {seconds * 0.4:.0f}s, {cores} threads; "
                   f"fp32-only: {mpix / t_f32:.1f} Mpix/s ({n_f32} calls)"
                   + (f"; 1 thread: {mpix / t_one:.1f} Mpix/s" if t_one else "")),
        "us_per_image": round(t_u8 * 1e6, 1), "f32_value": round(mpix / t_f32, 2),
        "value_1_thread": round(mpix / t_one, 2) if t_one else None,
    }


def secondary_configs(dev, with_atomics=False):
    """Short timings (rank 0, N=1, after the headline's timed region) of the other BASELINE.json configs, so that every
    round's BENCH json carries them: algorithmic GB/s = (input + output bytes) / event time.  Not the headline metric."""
    from interpolate_antialiasing_amd import _lib, extension_interpolate as aa

    def timed(fn, reps=50, warm_seconds=0.1):
        # Warm-up by TIME, not by count: for the first 40-60 ms of a new heavy workload the chip runs it 10-35 % slower than it then
        # settles to (per-launch durations of 600 back-to-back launches under rocprofv3: profiles/r03_launch_transient.txt) — 10 warm-up
        # launches of a 0.3 ms kernel sit inside that transient.  The headline's --prewarm-seconds serves the same purpose.
        for _ in range(10):
            fn()
        t_end = time.perf_counter() + warm_seconds
        while time.perf_counter() < t_end:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    res = []
    torch.manual_seed(1)

    def add(name, fn, nbytes):
        try:
            ms = timed(fn)
            res.append({"workload": name, "ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1),
                        "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4), "variant": _lib.last_variant()})
        except Exception as e:  # a secondary line must never take the headline down with it
            res.append({"workload": name, "error": str(e)[:200]})
        torch.cuda.empty_cache()

    x = torch.rand(256, 3, 438, 906, device=dev) * 255
    add("configs[0] on GPU: fp32 NCHW [256,3,438,906]->[196,320] bilinear", lambda: aa.linear_forward(x, [196, 320]),
        256 * 3 * 4 * (438 * 906 + 196 * 320))
    xd = x[:128].double()
    add("fp64 NCHW [128,3,438,906]->[196,320] bilinear (the reference dispatches double too)", lambda: aa.linear_forward(xd, [196, 320]),
        128 * 3 * 8 * (438 * 906 + 196 * 320))
    del xd
    x = x.contiguous(memory_format=torch.channels_last)
    add("fp32 channels_last [256,3,438,906]->[196,320] bilinear", lambda: aa.linear_forward(x, [196, 320]),
        256 * 3 * 4 * (438 * 906 + 196 * 320))
    del x
    x = torch.rand(64, 3, 1024, 1024, device=dev) * 255
    add("configs[2]: fp32 NCHW [64,3,1024,1024]->[224,224] bicubic", lambda: aa.cubic_forward(x, [224, 224]),
        64 * 3 * 4 * (1024 * 1024 + 224 * 224))
    add("configs[2] in the opt-in tolerance mode (precision='fast': FMA accumulation, <= 1e-4 relative; the line above is bit-exact)",
        lambda: aa.cubic_forward(x, [224, 224], precision="fast"), 64 * 3 * 4 * (1024 * 1024 + 224 * 224))
    xh = x.half()
    add("fp16 NCHW [64,3,1024,1024]->[224,224] bilinear, bit-exact half(reference_fp32)", lambda: aa.linear_forward(xh, [224, 224]),
        64 * 3 * 2 * (1024 * 1024 + 224 * 224))
    add("fp16 NCHW [64,3,1024,1024]->[224,224] bilinear, tolerance mode", lambda: aa.linear_forward(xh, [224, 224], precision="fast"),
        64 * 3 * 2 * (1024 * 1024 + 224 * 224))
    del x, xh
    x = torch.randint(0, 256, (1024, 906, 438, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
    add("configs[3] per-GPU shard: uint8 channels_last [1024,3,906,438]->[320,196] bilinear", lambda: aa.linear_forward(x, [320, 196]),
        1024 * 3 * (906 * 438 + 320 * 196))
    add("configs[3] shard in the reference harness's uint8 semantics (float(), fp32 op, truncating byte())",
        lambda: aa.linear_forward(x, [320, 196], uint8_mode="harness"), 1024 * 3 * (906 * 438 + 320 * 196))
    add("decode-adjacent: uint8 HWC [1024,906,438,3] -> float32 NCHW [1024,3,320,196] in one launch (+ mean/std)",
        lambda: aa.linear_forward(x, [320, 196], out_dtype=torch.float32, out_format="nchw", mean=[123.675, 116.28, 103.53],
                                  std=[58.395, 57.12, 57.375]), 1024 * 3 * (906 * 438 + 4 * 320 * 196))
    add("decode-adjacent in the opt-in tolerance mode (precision='fast': FMAs in both passes, float32 output within 1e-4 relative)",
        lambda: aa.linear_forward(x, [320, 196], out_dtype=torch.float32, out_format="nchw", mean=[123.675, 116.28, 103.53],
                                  std=[58.395, 57.12, 57.375], precision="fast"), 1024 * 3 * (906 * 438 + 4 * 320 * 196))
    x = x.contiguous()
    add("uint8 NCHW (planar) [1024,3,906,438]->[320,196] bilinear", lambda: aa.linear_forward(x, [320, 196]),
        1024 * 3 * (906 * 438 + 320 * 196))
    add("test.py's own uint8 path: CHW bytes [1024,3,906,438] in the harness's float arithmetic (float(), fp32 op, byte())",
        lambda: aa.linear_forward(x, [320, 196], uint8_mode="harness"), 1024 * 3 * (906 * 438 + 320 * 196))
    del x
    x = torch.randint(0, 256, (32, 2160, 3840, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
    add("thumbnails: uint8 channels_last [32,3,2160,3840]->[224,224] bicubic (69 taps: four lanes per pixel, DPP reduction; Pillow-exact)",
        lambda: aa.cubic_forward(x, [224, 224]), 32 * 3 * (2160 * 3840 + 224 * 224))
    del x
    xh = (torch.rand(64, 3, 438, 906, device=dev) * 255).half()
    add("fp16 NCHW up-scaling [64,3,438,906]->[1200,1200] bilinear, bit-exact half(reference_fp32)", lambda: aa.linear_forward(xh, [1200, 1200]),
        64 * 3 * 2 * (438 * 906 + 1200 * 1200))
    del xh
    g = torch.randn(256, 3, 196, 320, device=dev)
    add("configs[4] batched: backward fp32 grad [256,3,196,320]->[256,3,438,906], gather form (true adjoint)",
        lambda: aa.linear_backward(g, [196, 320], [256, 3, 438, 906]), 256 * 3 * 4 * (438 * 906 + 196 * 320))
    if with_atomics:  # the scatter-add form is kept for API parity with BASELINE configs[4]'s wording only (two memsets, an HBM intermediate,
        # one atomic per tap: 7 ms, 40x the gather form above); it is not a product path, so the default line does not carry it
        add("configs[4] batched, scatter-add ATOMICS form (API parity only; --with-atomics)",
            lambda: aa.linear_backward(g, [196, 320], [256, 3, 438, 906], atomic=True), 256 * 3 * 4 * (438 * 906 + 196 * 320))
    g1 = torch.randn(1, 3, 196, 320, device=dev)
    add("configs[4] as written: backward fp32 grad [1,3,196,320]->[1,3,438,906], gather form (latency)",
        lambda: aa.linear_backward(g1, [196, 320], [1, 3, 438, 906]), 3 * 4 * (438 * 906 + 196 * 320))
    x1 = torch.randint(0, 256, (1, 438, 906, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
    add("configs[1] as written: uint8 channels_last [1,3,438,906]->[196,320] (latency)", lambda: aa.linear_forward(x1, [196, 320]),
        3 * (438 * 906 + 196 * 320))
    try:  # cold call ("report cold first-call separately"): a shape never seen in this process — both weight tables are
        # built on device (one 64-byte header read-back each), then the kernel runs; wall clock around call + synchronize
        colds = []
        for k in range(5):
            xc = torch.randint(0, 256, (1, 430 + k, 900 + k, 3), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            aa.linear_forward(xc, [196, 320])
            torch.cuda.synchronize()
            colds.append((time.perf_counter() - t0) * 1e3)
        res.append({"workload": "cold call: a new shape [1,3,430+k,900+k]->[196,320] uint8 channels_last, two device-side table builds + "
                                "launch, wall clock incl. synchronize (median of 5 shapes)", "ms": round(sorted(colds)[2], 4),
                    "variant": _lib.last_variant()})
    except Exception as e:
        res.append({"workload": "cold call (new shape)", "error": str(e)[:200]})
    try:  # the same B=1 call replayed from a HIP graph (what a latency-bound server would do)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            aa.linear_forward(x1, [196, 320])
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            aa.linear_forward(x1, [196, 320])
        ms = timed(graph.replay, reps=200)
        res.append({"workload": "configs[1] as written, HIP-graph replay (not a speed-up: hipGraphLaunch itself costs 10-16 us)", "ms": round(ms, 4),
                    "GB/s": round(3 * (438 * 906 + 196 * 320) / ms / 1e6, 1), "variant": "fused_u8_nhwc_pil_v3 (graph)"})
    except Exception as e:
        res.append({"workload": "configs[1] as written, HIP-graph replay (latency)", "error": str(e)[:200]})
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-seconds", type=float, default=0.3,
                    help="untimed spin-up before the W warm-up steps: the GPU needs ~20 ms of work to reach its sustained clock "
                         "(a 5-step warm-up under-reports the steady state by 8 %%, see DESIGN.md section 5)")
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short timings of the other BASELINE configs")
    ap.add_argument("--with-atomics", action="store_true", help="secondary: also time the scatter-add atomics form of the backward (API parity path)")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="exercise the rank launcher only: every rank prints its rendezvous environment and exits before any "
                         "GPU call; rank 0 also prints the bench line's skeleton (n_gpus, global_batch).  Runs without a GPU.")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched ranks (0 = pick a free one)")
    ap.add_argument("--force-launcher", action="store_true",
                    help="take the self-launching path even with --gpus 1: the parent starts ONE fresh rank process, which joins a "
                         "1-rank RCCL process group (init_process_group, table broadcast, barriers, reductions all run).  This is how the "
                         "multi-rank code path is exercised end to end on a one-GPU box.")
    ap.add_argument("--rank-timeout", type=float, default=1500.0, help="launcher: seconds before the ranks are killed")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed.run around it: this parent, which has NOT touched
    the GPU (no torch.cuda call, no HIP call: importing torch does not initialise it), starts N fresh rank processes of
    this same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits for all of them, forwards
    rank 0's stdout (the single JSON line) and exits non-zero if any rank did.  Nothing is exec'ed over a live process."""
    import socket
    import subprocess

    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    import tempfile
    import threading

    procs, errfiles = [], []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if args.force_launcher:
            env["AA_BENCH_FORCE_DIST"] = "1"
        # stderr of ranks > 0 goes to a temporary FILE (a chatty rank — RCCL debug output, a traceback storm — can then never block
        # on a full pipe while rank 0 waits for it in a collective); every rank's stdout is drained by its own thread
        ef = None if rank == 0 else tempfile.TemporaryFile(mode="w+")
        errfiles.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--force-launcher"], env=env,
                                      stdout=subprocess.PIPE, stderr=ef, text=True))
    outs = [""] * len(procs)

    def drain(i):
        outs[i] = procs[i].stdout.read()

    threads = [threading.Thread(target=drain, args=(i,), daemon=True) for i in range(len(procs))]
    for t in threads:
        t.start()
    deadline = time.time() + args.rank_timeout
    rc, failed = 0, None
    while True:  # poll all ranks: when one fails (or time runs out) the others are killed instead of waiting in a collective for ever
        states = [p.poll() for p in procs]
        bad = [i for i, st in enumerate(states) if st not in (None, 0)]
        if bad or time.time() > deadline:
            failed = bad[0] if bad else -1
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        if all(st == 0 for st in states):
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    for t in threads:
        t.join(timeout=5)
    if failed is not None:
        rc = 1
        for rank, p in enumerate(procs):
            err = ""
            if errfiles[rank] is not None:
                errfiles[rank].seek(0)
                err = errfiles[rank].read()
            sys.stderr.write(f"[bench launcher] rank {rank} exited with {p.returncode}" + (" (first failure)" if rank == failed else "")
                             + (" [timeout]" if failed == -1 else "") + f"\n{err[-2000:]}\n")
    if args.launch_dry_run:  # every rank's environment line, rank order
        for out in outs:
            sys.stdout.write(out)
    else:
        sys.stdout.write(outs[0])
    sys.stdout.flush()
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_launcher):
        raise SystemExit(launch_ranks(args, argv))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and args.gpus != 1:  # (--gpus left at its default under torchrun: the environment decides)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with `python bench.py --gpus N` (self-launching) "
                         f"or `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N`")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.launch_dry_run:  # before any GPU call
        hook = os.environ.get("AA_BENCH_DRYRUN_HOOK", "")  # launcher tests: "chatty" = every rank > 0 writes 1 MB to stderr first;
        if hook == "chatty" and rank > 0:                  # "fail1" = rank 1 exits non-zero while rank 0 would wait for ever
            sys.stderr.write("x" * (1 << 20))
            sys.stderr.flush()
        if hook == "fail1":
            if rank == 1:
                raise SystemExit(3)
            time.sleep(600)
        line = {"dry_run": True, "rank": rank, "local_rank": local_rank, "world": world,
                "master": f"{os.environ.get('MASTER_ADDR', '')}:{os.environ.get('MASTER_PORT', '')}"}
        if rank == 0:
            line.update({"n_gpus": world, "config": {"batch_per_gpu": args.batch, "global_batch": args.batch * world,
                                                     "parallelism": f"batch-shard x{world}"}, "scaling": "weak"})
        print(json.dumps(line), flush=True)
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # (AA_BENCH_FORCE_DIST: set by --force-launcher for its one rank, so that a 1-rank RCCL group runs the whole multi-rank path)
    use_dist = world > 1 or os.environ.get("AA_BENCH_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from interpolate_antialiasing_amd import _lib, sharding
    from interpolate_antialiasing_amd import extension_interpolate as aa

    # one collective: rank 0's packed weight tables -> everyone (outside the timed region)
    sharding.prepare_tables(_lib.FILTER_LINEAR, _lib.TABLE_PIL, (H_IN, W_IN), (H_OUT, W_OUT), False, dev)

    B = args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(rank)
    # channels_last storage [B,H,W,C] viewed as NCHW; generated on device, resident in HBM before timing
    x = torch.randint(0, 256, (B, H_IN, W_IN, CH), dtype=torch.uint8, device=dev, generator=gen).permute(0, 3, 1, 2)
    assert x.is_contiguous(memory_format=torch.channels_last)

    def step():
        return aa.linear_forward(x, [H_OUT, W_OUT], False)

    t_spin = time.perf_counter() + max(0.0, args.prewarm_seconds)  # clock ramp: untimed, same work as a step
    while time.perf_counter() < t_spin:
        for _ in range(20):
            y = step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        y = step()
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # on the current stream = the stream the shim hands to the C-ABI
    for _ in range(args.steps):
        y = step()
    ev1.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    wall = sharding.reduce_max_seconds(wall, dev)
    total_images = sharding.reduce_sum_int(B * args.steps, dev)

    if rank == 0:
        # parity spot-check of what was just timed (image 0 of rank 0) against the oracle
        import oracle

        max_abs_e = 0
        for i in sorted({0, B // 2, B - 1}):  # first, middle and last image of the batch that was timed
            exp = oracle.pil_resize_u8("linear", x[i:i + 1].cpu().numpy(), (H_OUT, W_OUT))
            max_abs_e = max(max_abs_e, int(np.abs(y[i:i + 1].cpu().numpy().astype(int) - exp.astype(int)).max()))
        mpix_s = total_images * H_IN * W_IN / wall / 1e6
        alg_bytes_img = CH * H_IN * W_IN + CH * H_OUT * W_OUT  # 1,378,644
        kern_ms = ev_ms / args.steps  # one hot-path pass per step
        achieved = alg_bytes_img * B / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src, traffic_stale = pmc_traffic_bytes(variant, B)
        in_bytes_img = CH * H_IN * W_IN
        del y
        ceiling, ceiling_forms = measure_copy_ceiling(dev) if world == 1 else (None, {})  # (after the timed region; N = 1 only)
        out = {
            "metric": "Mpix/s (input pixels) antialiased bilinear 438x906->196x320, uint8 channels_last, PIL-exact",
            "value": round(mpix_s, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"uint8 channels_last [{B},3,438,906]->[196,320] bilinear antialias per GPU "
                                   f"(BASELINE configs[1] batched), Pillow-exact integer arithmetic",
                       "batch_per_gpu": B, "global_batch": B * world, "variant": variant, "prewarm_s": args.prewarm_seconds,
                       "parallelism": f"batch-shard x{world}", "lib": os.path.relpath(_lib.LIB_PATH, ROOT),
                       "kernel_sources": _lib.source_fingerprint(variant), "process_group": "nccl" if use_dist else None},
            "max_abs_err_vs_oracle": max_abs_e,
            "images_per_s": round(total_images / wall, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # north_star's wording is the HBM-READ roofline: input bytes only
                         "read_frac": round(in_bytes_img * B / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "kernel": variant, "kernel_ms": round(kern_ms, 4), "alg_bytes_per_launch": alg_bytes_img * B,
                         "copy_ceiling_measured_GBs": None if ceiling is None else round(ceiling, 1),
                         "copy_ceiling_by_kernel_form": ceiling_forms,
                         "frac_of_measured_copy_ceiling": None if ceiling is None else round(achieved / ceiling, 4),
                         "frac_of_guide_copy_ceiling_6290": round(achieved / GUIDE_COPY_CEILING_GBS, 4)},
        }
        # secondary workloads, the copy ceiling and the CPU baseline are rank-0-at-N=1 measurements (the contract): an N > 1 line says
        # so explicitly instead of dropping the keys
        if world > 1:
            out["secondary"], out["cpu_baseline"] = None, None
            out["n1_only"] = "secondary, cpu_baseline and the copy ceiling are measured at --gpus 1 only"
        else:
            out["secondary"] = None if args.no_secondary else secondary_configs(dev, args.with_atomics)
            out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
