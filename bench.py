#!/usr/bin/env python3
"""bench.py — VLP-16 scans/sec of the per-scan detector/descriptor hot path on MI355X.

A "step" is one pass of the whole hot path (rotate+filter -> per-ring clustering -> merge ->
3DSC descriptors, ref: src/feature_extraction_node.cpp:83-115) over one batch of synthetic
scans that are already resident in HBM, plus — at N > 1 — the one RCCL collective of the
path (all-gather — or, --gather root, gather — of each rank's compact keypoint block).  Scans are frame-sharded: every rank
owns `--batch` scans (weak scaling); there is no other data-path exchange.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant
kernel, HIP-event timed on the launch stream inside the timed region), `cpu_baseline`
(the oracle's kd-tree restatement of the PCL path on the host cores; N=1 only), and — N=1 only,
after the timed region — `other_configs` (BASELINE.json configs 3 and 5 at their stated batch
sizes) and `host_to_host_scans_per_s` (the same batch handed over and returned as host buffers).
"""
import argparse
import concurrent.futures as cf
import json
import os
import sys
import threading
import time

# HIP streams beyond the runtime's default of 4 hardware queues share queues (and then run one after the other): the batches
# in flight each need their own, next to torch's and RCCL's.  Read by the HIP runtime when it starts, so set before
# anything loads it; an explicit setting in the environment wins.  (a host application does the same.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured achievable)
N_RINGS, N_AZ = 16, 1800

# BASELINE.json configs 3 and 5 (n_rings / secondary_max beyond 16 are the build's extension:
# the reference hard-codes 16, ref: node.cpp:195, 200, 227).  Config 5 runs the launch preset SURVEY.md B-6 sized it with.
OTHER_CONFIGS = {
    # the headline workload under the node's constructor defaults (ref: node.cpp:9-34) instead of the launch file's preset
    "config2_vlp16_default_preset_batch1024": dict(
        batch=1024, n_uniq=64, synth=dict(), preset="default", params=dict(), limits=dict()),
    "config3_hdl64_64x2048_batch256": dict(
        batch=256, n_uniq=16, synth=dict(n_rings=64, n_az=2048, el0_deg=-24.8, el_step_deg=26.8 / 63, n_poles=256),
        preset="launch", params=dict(n_rings=64, el0_deg=-24.8, el_step_deg=26.8 / 63, secondary_max=64),
        limits=dict(max_candidates=4096, max_kpc_points=32768, max_keypoints=512, max_total_keypoints=256 * 256)),
    "config5_dense_128x2048_R2m_batch64": dict(
        batch=64, n_uniq=8, synth=dict(n_rings=128, n_az=2048, el0_deg=-25.0, el_step_deg=40.0 / 127, n_poles=256),
        preset="launch", params=dict(n_rings=128, el0_deg=-25.0, el_step_deg=40.0 / 127, secondary_max=128, descriptor_radius=2.0),
        limits=dict(max_candidates=8192, max_kpc_points=65536, max_keypoints=512, max_total_keypoints=64 * 256),
        in_flight=4),  # (measured 2 / 3 / 4 / 5 / 6 / 8 in flight: 5.9 / 7.5 / 8.5 / 8.1 / 8.5 / 7.5e4 — four since the streaming pass and the ring split
                       #  take several workgroups a scan; before that six were needed for 8.1e4)
}


def make_scans(capi, seeds, threads, **over):
    def one(seed):
        return capi.synth_scan(capi.synth_cfg(seed, **over))
    with cf.ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(one, seeds))


def cpu_baseline(params, host, roll, pitch, threads, budget_s=12.0):
    """Oracle (kd-tree search: the CPU restatement of the PCL path) on all host cores, native threads,
    one scan per thread at a time, the batch's first scans cycled for about `budget_s` seconds."""
    from oracle import oracle_py as O
    sample = host[:min(len(host), 256)]
    rate, n, kps, dt = O.bench_throughput(params, sample, roll, pitch, threads, budget_s)
    return {"value": rate, "unit": "scans/s", "cores": threads, "kind": "port",
            "sample": f"{n} scans in {dt:.1f} s (the batch's first {len(sample)} scans cycled): oracle/fx_oracle.cpp, the CPU "
                      f"restatement of the PCL path with its own kd-tree (leaf 15), g++ -O2, {threads} native threads, one "
                      f"scan per thread at a time, {kps} keypoints; PCL itself cannot be installed here"}


def h2d_inclusive(ctx, capi, host, B, N, roll, pitch, steps=3):
    """Same batch handed over as HOST buffers (pageable numpy): the C-ABI copies it to the device
    inside the call.  Results stay on the device.  Reported beside the headline number, never as it."""
    descs = ctx.make_descs([host.ctypes.data + b * N * 16 for b in range(B)], [N] * B, 16, roll, pitch)
    ctx.process_raw(descs, B, 0)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.process_raw(descs, B, 0)
    ctx.synchronize()
    return B * steps / (time.perf_counter() - t0)


def host_to_host(ctxs, capi, torch, host, B, N, roll, pitch, steps=4):
    """What a caller of the node actually receives (ref: node.cpp:117-139): scans go in as host buffers and keypoints +
    descriptors come back in (pinned) host buffers — H2D, every kernel and D2H inside fx_process_batch(FX_OUT_HOST).
    One host thread per context (the C-ABI's threading model: one context per thread), each on its own stream, so the
    upload of one batch, the kernels of another and the download of a third overlap; PCIe is full duplex.
    The input lives in pinned host memory (what a driver that feeds a GPU would allocate)."""
    pinned = torch.from_numpy(host).pin_memory()
    base = pinned.data_ptr()
    descs = ctxs[0].make_descs([base + b * N * 16 for b in range(B)], [N] * B, 16, roll, pitch)
    for c in ctxs:  # first call allocates the pinned mirrors
        c.process_raw(descs, B, capi.FX_OUT_HOST)
    err = []

    def worker(c):
        try:
            for _ in range(steps):
                c.process_raw(descs, B, capi.FX_OUT_HOST)
        except Exception as e:  # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    if err:
        raise err[0]
    return B * steps * len(ctxs) / dt


def run_other_config(name, cfg, capi, torch, dev, threads, roll, pitch, steps=6, in_flight=4):
    """One of BASELINE.json's other configurations at its stated batch size: scans/s with inputs resident in HBM,
    capacity flags, keypoints per scan.  `in_flight` contexts take the steps in turn, each on its own HIP stream — the
    way the headline number is measured (a batch of these configurations is a few hundred workgroups per kernel: one
    batch alone leaves most of the GPU idle); `one_at_a_time` is the same measurement with a single context.
    B distinct device buffers (n_uniq scenes cycled)."""
    B, n_uniq = cfg["batch"], cfg["n_uniq"]
    uniq = make_scans(capi, [10 + b for b in range(n_uniq)], min(threads, n_uniq), **cfg["synth"])
    N = len(uniq[0])
    d_in = torch.from_numpy(np.stack([uniq[b % n_uniq] for b in range(B)])).to(dev)
    p = capi.params(cfg["preset"], **cfg["params"])
    ctxs = [capi.Context(p, capi.limits(B, N, **cfg["limits"]), device=dev.index) for _ in range(in_flight)]
    ctx = ctxs[0]
    descs = ctx.make_descs([d_in.data_ptr() + b * N * 16 for b in range(B)], [N] * B, 16, roll, pitch)
    # the same scans one place on: the steps alternate between the two orders, so that a context never gets the batch it
    # had last time (its descriptor rows keep their content between batches)
    descs_b = ctx.make_descs([d_in.data_ptr() + ((b + 1) % B) * N * 16 for b in range(B)], [N] * B, 16, roll, pitch)
    torch.cuda.synchronize(dev)

    def timed(cs, n_steps):
        for c in cs:  # (launch-policy hint: how many of the caller's batches share the chip — grids and launch shapes, never results)
            c.set_batches_in_flight(len(cs))
        for j in range(2 * len(cs)):
            cs[j % len(cs)].process_raw(descs_b if (j // len(cs)) % 2 else descs, B, capi.FX_IN_DEVICE)
        for c in cs:
            c.synchronize()
        t0 = time.perf_counter()
        for j in range(n_steps):
            cs[j % len(cs)].process_raw(descs_b if (j // len(cs)) % 2 else descs, B, capi.FX_IN_DEVICE)
        for c in cs:
            c.synchronize()
        return (time.perf_counter() - t0) / n_steps

    dt1 = timed(ctxs[:1], steps)
    dt = timed(ctxs, steps * in_flight)
    ctx.set_profiling(3)
    for _ in range(3):
        ctx.process_raw(descs, B, capi.FX_IN_DEVICE)
    ctx.synchronize()
    acc = {}
    for back in range(3):
        ms, _tot = ctx.timings(back)
        for k, v in ms.items():
            acc[k] = acc.get(k, 0.0) + v / 3
    ctx.set_profiling(0)
    flags_or, k_total = 0, 0
    for c in ctxs:  # every context's results: the same batch, so the same counts, and no flags on any of them
        v = c.process_raw(descs, B, capi.FX_IN_DEVICE | capi.FX_OUT_HOST)
        flags_or |= int(np.bitwise_or.reduce(np.ctypeslib.as_array(v.h_flags, shape=(B,))))
        k_total = int(v.total_keypoints)
    alg = 16.0 * N * B + (16.0 + 7956.0) * k_total
    out = {"scans_per_s": B / dt, "ms_per_batch": dt * 1e3, "batches_in_flight": in_flight,
           "one_at_a_time": {"scans_per_s": B / dt1, "ms_per_batch": dt1 * 1e3},
           "batch": B, "points_per_scan": N,
           "preset": cfg["preset"], "flags_or": flags_or, "keypoints_per_scan": k_total / B,
           "alg_bytes_per_batch": alg, "path_frac_of_hbm_peak": alg / dt / 1e9 / HBM_PEAK_GBS,
           "kernel_ms": {k: round(x, 4) for k, x in acc.items()},
           "kernel_ms_source": "one batch at a time, HIP events around every stage"}
    for c in ctxs:
        c.close()
    del d_in
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="scans per GPU per step")
    ap.add_argument("--preset", default="launch", choices=["default", "launch"])
    ap.add_argument("--contexts", type=int, default=4,
                    help="batches in flight per GPU: contexts (each on its own HIP stream) taking the steps in turn")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-descriptors", action="store_true", help="diagnostic: estimate_descriptors = 0 (the detector alone; not the headline metric)")
    ap.add_argument("--no-extras", action="store_true", help="skip other_configs and the host-to-host measurement")
    ap.add_argument("--check", type=int, default=16, help="scans of rank 0 checked against the oracle after timing")
    ap.add_argument("--gather", default="all", choices=["all", "root"],
                    help="the path's one collective: every rank gets the keypoint table (ncclAllGather) or rank 0 only (ncclGather)")
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of --steps steps (0: as many as fill --target-seconds); the median is reported")
    ap.add_argument("--target-seconds", type=float, default=1.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from feature_extraction_amd import bench_dist, build, capi, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one rank builds (a stale library would otherwise be rewritten by every rank at once); the others wait for it.  The
    # distributed scaffolding — this handshake, the agreement on RCCL's C API, the kernel every rank times, the timed region
    # and its repeat count, the step loop — lives in feature_extraction_amd/bench_dist.py, where a gloo world-2 test runs it.
    bench_dist.wait_for_library(local_rank, build.build, build.stale, log=lambda m: print(m, file=sys.stderr))
    capi.load()  # raises if the HIP library is missing
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # FX_BENCH_ONE_DEVICE=1 (TEST mode, tests/test_gpu_bench_multi.py): every rank on device 0 over gloo, the gather through
    # host memory — the N > 1 control flow of this script (launch environment, build handshake, seeds, agreement, the timed
    # regions, rank 0's checks and its one JSON line) on a box with ONE GPU.  Never a measurement: RCCL refuses two ranks
    # on one device, and the ranks share the chip.
    one_device = os.environ.get("FX_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("FX_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if rank == 0:
            print(f"[bench] torch.distributed backend {dist.get_backend()} (RCCL), world {dist.get_world_size()}", file=sys.stderr)
    coll = bench_dist.Collectives(torch, dist, "cpu" if one_device else dev, rank, world, use_dist)

    B, N = args.batch, N_RINGS * N_AZ
    roll, pitch = 0.02, -0.015
    threads = os.cpu_count() or 1
    # ---- synthetic input (SURVEY.md Appendix C / BASELINE.md config 2), seed 1000 + global scan index
    seeds = bench_dist.rank_seeds(rank, world, B)
    scans = make_scans(capi, seeds, max(1, threads // max(1, min(world, 8))))
    host = np.stack(scans)  # [B, N, 4]
    d_in = torch.from_numpy(host).to(dev)  # resident in HBM before the timed region
    params = capi.params(args.preset, estimate_descriptors=0) if args.no_descriptors else capi.params(args.preset)
    # K contexts, each on its own HIP stream, take the steps in turn: the stage kernels of a batch are latency
    # bound and leave issue slots, LDS and whole CUs idle (tails, the large tiers' thin launches), which the
    # kernels of the other batches in flight fill.  One step is still one pass over one batch of B scans.
    K = max(1, args.contexts)
    ctxs = [capi.Context(params, capi.limits(B, N, sparse=True), device=dev_index) for _ in range(K)]  # (fx_limits_sparse: VLP-16 scans never need the dense tier's full pools: 4.5 GB a context, not 7)
    ctx = ctxs[0]
    for c in ctxs:
        c.set_batches_in_flight(K)  # (launch-policy hint: the grid-stride kernels share the chip with the other batches' kernels)
    # What crosses GPUs: ONE compact keypoint block per rank and batch (fx_pack_keypoint_block: offsets + flags + the keypoints
    # packed in scan order), 64 keypoints a scan of capacity — 1.06 MB a rank and step where the fixed-stride records of 0.6
    # (stride = the context's capacity, 256) were 4.21 MB; a batch with more keypoints is cut and FLAGGED (asserted below).
    BLOCK_KP = sharding.block_keypoints_per_scan(B)
    BLOCK_ROWS = sharding.block_rows(B, BLOCK_KP)
    # the contexts' own HIP streams, wrapped for torch (streams from torch's pool can share a hardware queue: two such
    # contexts then do not overlap at all; FX_BENCH_TORCH_STREAMS=1 brings them back)
    if os.environ.get("FX_BENCH_TORCH_STREAMS") == "1":
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    else:
        streams = [torch.cuda.ExternalStream(c.stream_ptr(), device=dev) for c in ctxs]
    for c, st in zip(ctxs, streams):
        c.set_stream(st.cuda_stream)
    base = d_in.data_ptr()
    descs = ctx.make_descs([base + b * N * 16 for b in range(B)], [N] * B, 16, roll, pitch)
    # A second batch of different scans: the steps alternate between the two, so no context ever sees the batch it
    # processed last time (descriptor rows keep their content between batches — k_desc_group clears a row by un-writing
    # what it wrote — and a repeated batch would clear and rewrite the very same bins).
    scans_b = make_scans(capi, bench_dist.rank_seeds(rank, world, B, second=True), max(1, threads // max(1, min(world, 8))))
    d_in_b = torch.from_numpy(np.stack(scans_b)).to(dev)
    descs_b = ctx.make_descs([d_in_b.data_ptr() + b * N * 16 for b in range(B)], [N] * B, 16, roll, pitch)
    del scans_b
    # keypoint records, one buffer per context: the gather of a batch (RCCL, its own stream) overlaps the
    # kernels of the batches behind it; a buffer is reused only after its collective has completed
    recs = [torch.zeros((BLOCK_ROWS, 4), dtype=torch.float32, device=dev) for _ in range(K)]
    to_root = args.gather == "root"
    gathered = [torch.zeros((world * BLOCK_ROWS, 4), dtype=torch.float32, device=dev) for _ in range(K)] if use_dist and (rank == 0 or not to_root) else None
    # the collective goes straight on the context's stream through RCCL's C API (sharding.RcclGather);
    # FX_BENCH_TORCH_GATHER=1 takes torch.distributed's all_gather_into_tensor instead (sharding.all_gather_records, the
    # function the gloo test runs)
    rccl, rccl_comms = None, 1
    if use_dist and os.environ.get("FX_BENCH_TORCH_GATHER") != "1" and not one_device:
        # ONE communicator for all contexts, its collectives issued in step order on every rank: RCCL's ordering contract,
        # whatever the streams do.  Measured on one rank with the collective forced: 1.50 million scans/s against 1.68
        # without a collective — the shared communicator orders the contexts' streams against each other.  One
        # communicator per context (FX_BENCH_MULTI_COMM=1: 1.68; every rank issues them in the same host order, so no two
        # ranks can hold each other's kernels back, but that has never run on more than one GPU) and a dedicated gather
        # stream fed through events (1.30: cross-stream waits are dear on this stack) were measured too.
        # Every rank first proves it can load RCCL's C API; only then are ids exchanged.
        if coll.agree(sharding.RcclGather.available()):
            rccl_comms = K if os.environ.get("FX_BENCH_MULTI_COMM") == "1" else 1
            rccl = sharding.RcclGather(world, rank, dev, n_comms=rccl_comms)
        elif rank == 0:
            print("[bench] direct RCCL gather unavailable on some rank; using torch.distributed's all_gather", file=sys.stderr)
    torch.cuda.synchronize(dev)  # inputs and zeroed buffers are in place before any side stream starts

    def run_slot(j, which):
        ctxs[j].process_raw(descs_b if which else descs, B, capi.FX_IN_DEVICE)
        ctxs[j].pack_keypoint_block(recs[j].data_ptr(), B, BLOCK_KP)
        if rccl is not None and to_root:  # the path's one collective, an ordinary kernel of this context's stream
            rccl.gather(recs[j], gathered[j] if gathered is not None else None, streams[j].cuda_stream, root=0, comm=j % rccl_comms)
        elif rccl is not None:
            rccl.all_gather(recs[j], gathered[j], streams[j].cuda_stream, comm=j % rccl_comms)
        elif use_dist and one_device:  # (test mode: gloo has no device collective — through host memory, synchronously)
            streams[j].synchronize()
            if to_root:
                t = sharding.gather_records_to_root(recs[j].cpu(), world, root=0)
                if t is not None:
                    gathered[j].copy_(t)
            else:
                gathered[j].copy_(sharding.all_gather_records(recs[j].cpu(), world))
        elif use_dist and to_root:
            streams[j].synchronize()
            sharding.gather_records_to_root(recs[j], world, root=0, out=gathered[j].view(-1) if gathered is not None else None)
        elif use_dist:
            return sharding.all_gather_records(recs[j], world, out=gathered[j], async_op=True)[1]
        return None

    loop = bench_dist.StepLoop(K, run_slot, enter=lambda j: torch.cuda.stream(streams[j]))
    step, drain = loop.step, loop.drain

    def profile_all(depth, stages=None):
        for c in ctxs:
            c.set_profiling(depth, stages=stages)

    exec_span = {}  # k_prep's own execution span (device clock), mean of the batches the last mean_timings() read

    def mean_timings(per_ctx):
        acc, n, span, n_span = {}, 0, 0.0, 0
        for c in ctxs:
            for back in range(per_ctx):
                try:
                    ms, _tot = c.timings(back)
                except capi.FxError:  # fewer timed steps than contexts: this one has nothing that far back
                    break
                n += 1
                for k, v in ms.items():
                    acc[k] = acc.get(k, 0.0) + v
                if c.last_k_prep_exec_ms > 0:
                    span += c.last_k_prep_exec_ms
                    n_span += 1
        exec_span["k_prep"] = span / n_span if n_span else None
        return {k: v / n for k, v in acc.items()}

    # ---- warm-up (untimed; the first step of a context pays for code upload and cold caches)
    for _ in range(max(args.warmup, 2) * K):
        step()
    drain()
    torch.cuda.synchronize(dev)
    # ---- which kernel dominates: HIP events around every stage kernel over `n_sel` steps per context, the largest
    #      mean wins.  Every event costs a few microseconds of stream time (13 of them: ~4 % of a batch), so the
    #      timed region below keeps only the events that bracket that kernel (and the batch).
    n_sel = 5
    profile_all(n_sel)
    for _ in range(n_sel * K):
        step()
    drain()
    torch.cuda.synchronize(dev)
    stage_ms = mean_timings(n_sel)
    # The roofline line is about ONE kernel: the longest of the stages that are a single launch (a HIP-event span around
    # several launches is mostly queueing when other batches are in flight).
    # Several stages are within a few per cent of each other by now and the longest changes from run to run: among the
    # stages within 10 % of the longest, the one that moves the most algorithmic bytes is named — the roofline line is a
    # statement about bytes (k_rings_runs, 3 % of the HBM line whatever its duration, says nothing about the path).
    singles = {k: stage_ms[k] for k in capi.SINGLE_LAUNCH_STAGES}
    longest = max(singles, key=singles.get)
    pre_bytes = ctx.stage_bytes()
    close = [k for k in singles if singles[k] >= 0.9 * singles[longest]]
    dom = max(close, key=lambda k: sum(pre_bytes[k]))
    dom_rule = (f"of the single-launch stages within 10 % of the longest ({longest}: HIP-event mean over {n_sel} profiled steps per "
                f"context, {K} batches in flight, measured in this run before the timed region) the one with the most "
                "algorithmic bytes per launch")
    dom = capi.STAGE_NAMES[coll.broadcast_index(capi.STAGE_NAMES.index(dom))]  # every rank times the same kernel
    per_ctx_steps = max(1, args.steps // K)
    profile_all(per_ctx_steps, stages=[dom])  # on the launch streams, inside the timed region

    # EXACTLY args.steps steps between barrier + synchronize on both sides, the MAX over the ranks (bench_dist.timed_region).
    # A region of a few dozen steps lasts ~10 ms, of which filling and draining the batches in flight is a sixth, and a
    # driver that samples GPU activity never sees it: the region is repeated for about a second (the same count on every
    # rank: it follows from the first region's all-reduced time) and the MEDIAN region is reported; `steps`
    # stays the unit.
    regions = bench_dist.measure(args.steps, step, drain, lambda: torch.cuda.synchronize(dev), coll, repeats=args.repeats,
                                 target_seconds=args.target_seconds)
    elapsed = float(np.median(regions))

    # ---- the dominant kernel's duration over the timed steps (HIP events recorded inside the timed region)
    dom_ms = mean_timings(per_ctx_steps)[dom]
    dom_exec_ms = exec_span.get(dom)  # (k_prep only: first workgroup's start to last workgroup's end, device clock)
    profile_all(0)
    # ---- the same kernel with ONE batch on the chip (outside the timed region): what a kernel trace of `--contexts 1` shows
    #      (profiles/*_c1_kernel_stats.csv) — with batches in flight a launch's span stretches with whatever shares the chip
    n_alone = 20
    ctxs[0].set_profiling(n_alone, stages=[dom])
    with torch.cuda.stream(streams[0]):
        for i in range(n_alone):
            ctxs[0].process_raw(descs_b if i % 2 else descs, B, capi.FX_IN_DEVICE)
    torch.cuda.synchronize(dev)
    dom_alone_ms = float(np.mean([ctxs[0].timings(back)[0][dom] for back in range(n_alone)]))
    ctxs[0].set_profiling(0)

    # ---- what the batch produced (for the algorithmic byte count) + a parity spot check
    v = ctx.process_raw(descs_b, B, capi.FX_IN_DEVICE | capi.FX_OUT_HOST)
    k_total_b = int(v.total_keypoints)
    flags_or = int(np.bitwise_or.reduce(np.ctypeslib.as_array(v.h_flags, shape=(B,)))) if B else 0
    v = ctx.process_raw(descs, B, capi.FX_IN_DEVICE | capi.FX_OUT_HOST)
    k_total = (int(v.total_keypoints) + k_total_b) / 2.0  # keypoints per step: the two alternating batches' mean
    flags_or |= int(np.bitwise_or.reduce(np.ctypeslib.as_array(v.h_flags, shape=(B,)))) if B else 0
    stage_bytes = ctx.stage_bytes()  # algorithmic bytes (read, written) of every stage of that batch, from its own counts
    front = ctx.front_active()
    # non-zero descriptor values of that batch (a row of 1980 bins holds a dozen): what the descriptor stage physically has to
    # store — it writes those and un-writes the row's previous ones — where B_alg counts the whole 7956-byte row
    total_rows = int(v.total_keypoints)
    desc_nnz = 0
    if total_rows:
        rows = torch.from_numpy(np.ctypeslib.as_array(v.h_descriptors, shape=(total_rows * capi.FX_DESC_FLOATS,)))
        desc_nnz = int(torch.count_nonzero(rows).item())
    n_chk = min(args.check, B) if rank == 0 else 0
    res = []
    if n_chk:
        v2 = ctx.process_raw(descs, n_chk, capi.FX_IN_DEVICE | capi.FX_OUT_HOST | capi.FX_OUT_CLOUDS | capi.FX_OUT_DEBUG)
        res = ctx.unpack(v2)
    k_all = coll.sum(k_total)
    if use_dist and rank == 0:
        # the gathered table holds every rank's records in stream order: check this rank's block
        last = loop.last_slot
        g = gathered[last][rank * BLOCK_ROWS:(rank + 1) * BLOCK_ROWS]
        assert torch.equal(g, recs[last]), "gathered keypoint block differs from the local one"
        how = (("ncclGather to rank 0" if to_root else "ncclAllGather") + " on the context stream (RCCL)") if rccl is not None else dist.get_backend()
        print(f"[bench] {'gather' if to_root else 'all-gather'} of compact keypoint blocks over {how}: table {tuple(gathered[last].shape)} "
              f"({BLOCK_ROWS * 16 / 1e6:.2f} MB a rank), this rank's block equals its local records", file=sys.stderr)

    result_line = None
    if rank == 0:
        parity = None
        if n_chk:
            from oracle import oracle_py as O
            from tests import util
            worst, kchk = 0.0, 0
            with cf.ThreadPoolExecutor(max_workers=min(threads, n_chk)) as ex:  # (the oracle call releases the GIL)
                oras = list(ex.map(lambda b: O.run(params, scans[b], roll=roll, pitch=pitch), range(n_chk)))
            for b in range(n_chk):
                st = util.compare_scan(res[b], oras[b], tag=f"bench scan {b}")  # raises on any mismatch
                worst = max(worst, st["max_abs"])
                kchk += st["K"]
            parity = {"scans_checked": n_chk, "keypoints_checked": kchk, "keypoint_f1_vs_oracle": 1.0,
                      "cluster_membership": "exact", "descriptor_max_abs_diff": worst}
        # Whole path: 16 N read + 16 K + 7956 K written per scan; K measured, this rank's batch.
        path_bytes = 16.0 * N * B + (16.0 + 7956.0) * k_total
        # The named kernel's OWN algorithmic bytes per launch (fx_get_stage_bytes: what it must read of its inputs and write
        # of its outputs, once each, from the batch's counts): the bandwidth statement about that kernel.
        phys_bytes = (stage_bytes["k_prep"][0] + stage_bytes["k_prep"][1] + stage_bytes["k_merge"][1] + 8.0 * desc_nnz)
        # (stage 0 is ONE kernel either way: k_front — filter to keypoints — for scans that fit its LDS tables, else k_prep)
        dom_kernel = "k_front" if (dom == "k_prep" and front) else dom
        own_r, own_w = stage_bytes[dom]
        own_bytes = own_r + own_w
        achieved = own_bytes / (dom_alone_ms * 1e-3) / 1e9  # (one batch on the chip: the reading a kernel trace reproduces)
        ms_per_step = elapsed / args.steps * 1e3
        # HBM traffic is NOT measured in this run (PMC counters need rocprofv3 passes of their own): the figures are read
        # from profiles/traffic.json, written by tools/summarize_profile.py from the round's rocprofv3 --pmc passes over this
        # same command; `traffic_source` says which run that was.
        traffic = traffic_total = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # stage -> kernels launched inside it (PMC rows are per kernel)
                traffic = tj.get(dom_kernel) or sum(tj.get(k, 0.0) for k in capi.STAGE_KERNELS.get(dom, (dom,))) or None
                traffic_total = sum(x for k, x in tj.items() if k.startswith("k_"))
                traffic_source = "profiles/traffic.json, not measured in this run: " + str(tj.get("_source", "source not recorded"))
            except Exception:
                traffic = traffic_total = traffic_source = None
        flag_msg = None
        if use_dist:  # every rank's block header carries the OR of its scans' flags: a block that was cut (more keypoints than it holds) shows here
            last = loop.last_slot
            hdr = gathered[last].view(torch.int32).view(world, BLOCK_ROWS, 4)[:, 0, :].cpu().numpy()
            flag_msg = int(np.bitwise_or.reduce(hdr[:, 2])) if len(hdr) else 0
            assert flag_msg == 0, f"gathered keypoint blocks carry flags 0x{flag_msg:x} (0x4: more keypoints than the block holds)"
            assert all(int(h[0]) == B for h in hdr), "a gathered block does not hold the rank's whole batch"
        out = {
            "metric": "VLP-16 scans/sec (16x1800 pts), detector+descriptor", "value": world * B * args.steps / elapsed,
            "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            # the timed region (exactly `steps` steps between barriers) repeated; value / ms_per_step are the MEDIAN region's
            "repeats": len(regions), "region_ms": {"median": elapsed * 1e3, "min": min(regions) * 1e3, "max": max(regions) * 1e3},
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batch of {B} synthetic VLP-16 scans (16x1800 pts, 64 uniform poles) per GPU, "
                                   f"device-resident, preset '{args.preset}', roll/pitch 0.02/-0.015",
                       "scans_per_gpu": B, "points_per_scan": N, "preset": args.preset,
                       "parallelism": f"frame-sharded x{world}" + ((", gather to rank 0" if to_root else ", all-gather") + f" of compact keypoint blocks ({BLOCK_ROWS * 16 / 1e6:.2f} MB a rank, RCCL)" if world > 1 else "")
                                      + (" — TEST MODE: all ranks on one device over gloo, not a measurement" if one_device else ""),
                       "keypoints_per_scan": k_all / (world * B), "flags_or": flags_or, "batches_in_flight": K,
                       "gathered_record_flags_or": flag_msg},
            # frac = the dominant kernel's own algorithmic bytes / its duration with ONE batch on the chip (HIP events, 20
            # launches after the timed region): the reading a kernel trace of `bench.py --contexts 1` reproduces
            # (profiles/*_c1_kernel_stats.csv).  frac_exec = the same bytes / the kernel's execution span (device clock, first
            # workgroup's start to last workgroup's end) INSIDE the timed region, the other batches in flight sharing the chip:
            # what rocprofv3 reports for the headline run (profiles/*_kernel_stats.csv).
            "roofline": {"bound": "hbm", "kernel": dom_kernel,
                         # (the profiling slot the kernel's events live in: slots are named after the separate kernels — slot
                         #  "k_prep" is stage 0, which k_front occupies when the fused kernel runs)
                         "event_slot": dom, "event_slot_index": capi.STAGE_NAMES.index(dom),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # Which regime `frac` / `achieved` / `kernel_ms` describe: since round 5 the kernel's duration
                         # with ONE batch on the chip — what a kernel trace of `--contexts 1` reproduces — NOT the regime of the
                         # headline value (K batches in flight): that reading is `frac_in_timed_region` (= frac_exec) below.
                         # Rounds 1-4 quoted the in-flight HIP-event span here; compare across rounds by the explicit keys.
                         "frac_regime": "one batch on the chip (definition v2, rounds 5-6); the timed region's own figure: frac_in_timed_region",
                         "frac_one_at_a_time": achieved / HBM_PEAK_GBS, "kernel_ms_one_at_a_time": dom_alone_ms,
                         "frac_in_timed_region": (own_bytes / (dom_exec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom_exec_ms else None,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "definition": "achieved = this kernel's own algorithmic bytes per launch (alg_bytes_read + alg_bytes_written, "
                                       "fx_get_stage_bytes) / kernel_ms, its duration with one batch on the chip; frac_exec: the same "
                                       "over its execution span with the other batches in flight; path_frac = the whole path's "
                                       "algorithmic bytes per step (SURVEY.md 8d) / ms_per_step / peak: the figure to hold against "
                                       "north_star's 0.40",
                         "alg_bytes_per_launch": own_bytes, "alg_bytes_read": own_r, "alg_bytes_written": own_w,
                         "kernel_ms": dom_alone_ms,
                         "kernel_ms_source": f"HIP events around this kernel on the launch stream, {n_alone} launches of one context after the "
                                             "timed region, nothing else on the chip",
                         # inside the timed region, K batches in flight
                         "kernel_exec_ms": dom_exec_ms,
                         "kernel_exec_ms_source": f"device clock, first workgroup's start to last workgroup's end, inside the timed region ({K} batches in flight): what rocprofv3 reports",
                         "frac_exec": (own_bytes / (dom_exec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom_exec_ms else None,
                         # (the HIP-event span of a launch with batches in flight also counts its wait for free CUs behind the
                         #  other batches: a queueing figure, kept as a diagnostic only)
                         "kernel_event_span_ms_in_flight": dom_ms,
                         "selected_by": dom_rule,
                         "longest_stage": max(stage_ms, key=stage_ms.get),
                         "kernel_traffic_gbs": (traffic / (dom_alone_ms * 1e-3) / 1e9) if traffic else None,
                         "path_alg_bytes_per_step": path_bytes,
                         "path_frac": path_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # the task statement's literal reading (the per-scan figure x scans / ONE kernel's time): kept for
                         # comparison with earlier rounds, not a bandwidth statement about any kernel
                         "path_bytes_over_kernel_ms_frac": path_bytes / (dom_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic_total": traffic_total,
                         "traffic_over_path_alg_bytes": (traffic_total / path_bytes) if traffic_total else None,
                         # what physically has to move: the input once; every output the boundary exposes once (~cloud, near bits,
                         # keypoints_full + maps, keypoints, keypoint_cloud); the support lists' row tables; of the descriptor rows
                         # only their non-zero values, written and later un-written (8 bytes each) — B_alg counts whole rows
                         "physical_min_bytes": phys_bytes,
                         "traffic_over_physical_min": (traffic_total / phys_bytes) if traffic_total else None,
                         # every stage's own bytes and the fraction they make of the peak over the stage's pre-pass duration
                         "stage_alg_bytes": {k: [r, w] for k, (r, w) in stage_bytes.items()},
                         "stage_frac": {k: ((r + w) / (stage_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms.get(k) else None)
                                        for k, (r, w) in stage_bytes.items()}},
            "kernel_ms": stage_ms,
            "kernel_ms_source": f"all stages: {n_sel * K} profiled steps before the timed region, {K} batches in flight",
            "parity": parity,
        }
        result_line = out
    if rccl is not None:
        torch.cuda.synchronize(dev)
        rccl.close()
    for c in ctxs[1:] if (rank == 0 and world == 1 and not args.no_extras) else ctxs:
        c.close()
    if rank == 0 and world == 1:
        out = result_line
        if not args.no_extras:
            out["h2d_inclusive_scans_per_s"] = h2d_inclusive(ctx, capi, host, B, N, roll, pitch)
            ctx.close()
            h2h = [capi.Context(params, capi.limits(B, N, sparse=True), device=dev_index) for _ in range(3)]
            # (on the contexts' own streams: a torch stream that has run work keeps its hardware queue for the life of the process —
            #  three of eight — and the six contexts of config 5 below then shared the rest: 6.2e4 instead of 8.3e4 scans/s)
            out["host_to_host_scans_per_s"] = host_to_host(h2h, capi, torch, host, B, N, roll, pitch)
            out["host_to_host_note"] = ("pinned host scans in, keypoints + descriptors out to pinned host buffers "
                                        "(fx_process_batch with FX_OUT_HOST), 3 contexts on 3 host threads / streams")
            for c in h2h:
                c.close()
            del h2h
            del d_in
            torch.cuda.empty_cache()
            out["other_configs"] = {name: run_other_config(name, cfg, capi, torch, dev, threads, roll, pitch, in_flight=cfg.get("in_flight", 4))
                                    for name, cfg in OTHER_CONFIGS.items()}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, host, roll, pitch, threads)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if result_line is not None:
        # the one JSON line goes out last: RCCL writes a version banner through C stdio, which sits in
        # libc's buffer until flushed when stdout is a pipe
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result_line), flush=True)


if __name__ == "__main__":
    main()
