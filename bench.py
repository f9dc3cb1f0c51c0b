#!/usr/bin/env python3
"""Benchmark of the matching hot path.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2]

A "step" is one pass of the matcher over one batch: all Hungarian-matched output sets of
one training step (cfg2: B=64 images, 6 decoder-layer sets, Q=600, ~16 targets per image,
focal class cost) from "predictions and targets resident in HBM" to "CPU int64 index
tensors in the caller's hands" -- one `match_many` call: the target tensors checked and their
pointers collected, the step's kernel(s) (at C <= 4 ONE kernel: every assignment workgroup
computes its own cost block in LDS, reads the targets where they are and stores the indices
into the pinned host tensor the returned tuples are views of; at C > 4 a pack launch, the cost
kernel and the assignment kernel), the completion event, the per-image CPU tuples.
With N > 1 there is one process per GPU: started by torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE in the environment), or -- when `python bench.py --gpus N` is run on its own -- by
this script, which starts its N ranks as child processes before it touches a GPU.  Every rank
matches its own B images (`--scaling weak`, the default) or its share of the workload's B images,
dealt out by target count (`--scaling strong`), and the step also carries the single collective
of the path, the all-reduce of the matched-box counts (asynchronous, waited for after the kernels
have been enqueued).

Prints one JSON line: whole-job matched images/s, plus
  roofline      cost kernel on this run's inputs: algorithmic bytes per launch / HIP-event time per
                launch (back-to-back launches in a pass of its own), vs 8 TB/s
  per_layer_calls  the same step as the reference's unmodified criteria drive it: one matcher(output, targets)
                call per output set (detr/loss.py:164-169), each returning its CPU tuples before the next starts
  cpu_baseline  the reference's CPU formulation (torch ops on the whole batch + scipy per
                image, restated in oracle/torch_pipeline.py) timed on this host, rank 0, N=1.
"""
import argparse
import ctypes
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")


def measured_traffic(kernel, workload):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 counter passes
    (FETCH_SIZE doubled as gfx950 needs + WRITE_SIZE, separate --pmc runs of this script; made
    by tools/pmc_traffic.py).  None when the file does not cover this kernel at this workload."""
    try:
        with open(PMC_TRAFFIC) as f:
            table = json.load(f)
    except OSError:
        return None
    entry = table.get(kernel)
    if not entry or entry.get("workload") != workload:
        return None
    return entry["traffic_bytes"]


def cost_kernel_bytes(set_queries, n_logits, sizes):
    """Algorithmic bytes of one step's cost launches: logits + prediction boxes read,
    target boxes + int64 labels read once per problem, cost entries written."""
    total = 0
    for q in set_queries:
        for t in sizes:
            total += q * n_logits * 4 + q * 16 + t * (16 + 8) + q * t * 4
    return total


def cpu_baseline(wl, sets, targets, min_s=10.0, budget_s=20.0):
    """The reference's CPU formulation on this host: whole steps of the same workload, repeated
    until `min_s` seconds of CPU work are done (at most `budget_s`), at torch's default thread
    count; then a bounded single-thread sample of the same calls."""
    from oracle import torch_pipeline

    cpu_sets = [{k: v.cpu() for k, v in s.items()} for s in sets]
    cpu_targets = [{k: v.cpu() for k, v in t.items()} for t in targets]
    B = len(cpu_targets)
    # one untimed call to warm the allocator / thread pool
    torch_pipeline.match(cpu_sets[0]["pred_logits"][:2], cpu_sets[0]["pred_boxes"][:2], cpu_targets[:2],
                         **wl.matcher_kwargs)

    def run(min_time, max_time):
        done, t0 = 0, time.perf_counter()
        while True:
            s = cpu_sets[done % len(cpu_sets)]
            torch_pipeline.match(s["pred_logits"], s["pred_boxes"], cpu_targets, **wl.matcher_kwargs)
            done += 1
            t = time.perf_counter() - t0
            if t > max_time or (t > min_time and done % len(cpu_sets) == 0):
                return done, t

    done, elapsed = run(min_s, budget_s)
    step_s = elapsed * len(cpu_sets) / done
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        done1, elapsed1 = run(0.0, 6.0)
    finally:
        torch.set_num_threads(threads)
    return {
        "value": B / step_s, "unit": "images/s", "cores": threads, "kind": "port",
        "sample": f"{done} matcher calls ({done / len(cpu_sets):.2f} steps of {wl.name}, B={B}) in {elapsed:.1f} s; "
                  f"torch {torch.__version__} CPU ops on the whole-batch [B*Q, sum T] matrix + scipy per image",
        "single_thread_value": B / (elapsed1 * len(cpu_sets) / done1),
        "single_thread_sample": f"{done1} matcher calls in {elapsed1:.1f} s with torch.set_num_threads(1)",
    }


def stress_roofline(device, launches=20, workload="cfg4", copy_reference=True, given=None):
    """Cost kernel alone, timed with two events on the launch stream around `launches` back-to-back
    launches through the C ABI (no per-launch event bracket, no other work in between).
    Default: the cfg4 size (B=256, Q=900, C=91, heavy-tail T, 6 sets: 0.65 GB of algorithmic traffic
    per launch), the size at which an HBM-roofline fraction is meaningful."""
    import detr_amd
    from detr_amd import synth
    from detr_amd._lib import DmSet, lib
    from detr_amd.matcher import PackedTargets

    wl, sets, targets = given if given is not None else synth.make_workload(workload, device=device)
    workload = wl.name
    packed = PackedTargets(targets, device)
    S, M = len(sets), packed.total
    arr = (DmSet * S)()
    off = 0
    for s, o in enumerate(sets):
        q = o["pred_logits"].shape[1]
        stride = (q + 31) & ~31
        arr[s].logits, arr[s].boxes = o["pred_logits"].data_ptr(), o["pred_boxes"].data_ptr()
        arr[s].cost_off, arr[s].out_off, arr[s].Q, arr[s].q_stride = off, s * M, q, stride
        off += M * stride
    cost = torch.empty(off, dtype=torch.float32, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    prm = detr_amd.HungarianMatcher(**wl.matcher_kwargs)._params()
    stream = torch.cuda.current_stream(device).cuda_stream

    def launch():
        rc = lib.dm_cost_f32(arr, S, packed.batch, wl.n_logits, packed.labels.data_ptr(), packed.boxes.data_ptr(),
                             packed.offsets_ptr, packed.max_t, ctypes.byref(prm), cost.data_ptr(),
                             status.data_ptr(), stream)
        assert rc == 0, rc

    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        launch()
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / launches
    algo = cost_kernel_bytes(wl.set_queries, wl.n_logits, packed.sizes)
    achieved = algo / (ms * 1e-3) / 1e9
    del cost, sets
    kernel = "cost_small_kernel" if wl.n_logits <= 4 else "cost_tile_kernel"
    if not copy_reference:
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(kernel, workload) if packed.batch == wl.batch else None,
                "kernel": kernel, "algorithmic_bytes_per_launch": algo, "ms_per_launch": ms, "launches": launches,
                "what": f"dm_cost_f32 on this run's {workload} inputs: {launches} back-to-back launches between two "
                        "HIP events on the launch stream, in a pass of its own"}
    # what a plain streaming kernel reaches on this GPU: a 1 GiB device-to-device copy (read + write)
    a = torch.empty(1 << 28, dtype=torch.float32, device=device)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(device)
    copy_gbs = 2 * a.numel() * 4 / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9
    del a, b
    return {"workload": "cfg4 cost kernel only: B=256, Q=900, C=91, heavy-tail T, 6 sets", "bound": "hbm",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": measured_traffic("cost_tile_kernel", "cfg4"),
            "device_copy_gbs": copy_gbs, "frac_of_device_copy": achieved / copy_gbs,
            "kernel": "cost_tile_kernel", "algorithmic_bytes_per_launch": algo, "ms_per_launch": ms}


def flat_path(matcher, sets, targets, device, steps):
    """The same step through the opt-in interface that keeps everything on the device: targets
    packed once per batch (as a collate function would), `match_flat` (flat index tensors, no
    device-to-host copy, no per-image views).  Not the headline value: a different contract."""
    import detr_amd

    packed = detr_amd.PackedTargets(targets, device)
    for _ in range(10):
        matcher.match_flat(sets, packed)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        matcher.match_flat(sets, packed)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    matcher.check_status()
    return {"value": len(targets) * steps / elapsed, "unit": "images/s", "ms_per_step": 1e3 * elapsed / steps,
            "what": "match_flat on pre-packed targets: device-resident flat indices, one sync after all steps"}


def criterion_step(matcher, wl, sets, fresh, device, steps):
    """Not the headline: the matcher together with what consumes its indices -- matching, the L1 / GIoU losses of the
    matched pairs, the class maps and the classification loss of every output set, forward and backward, as one
    weighted scalar (detr_amd.losses.set_loss_total: the loop of /root/reference/models/detr/loss.py:164-176 and
    calc_total_loss :135-158 for one output group).  New target tensors every step; one synchronisation at the end."""
    from detr_amd import losses

    leaves = [dict(o, pred_logits=o["pred_logits"].detach().clone().requires_grad_(True),
                   pred_boxes=o["pred_boxes"].detach().clone().requires_grad_(True)) for o in sets]
    weight = torch.ones(wl.n_label_classes + 1, device=device)
    weight[-1] = 0.1

    def step(turn):
        total = losses.set_loss_total(matcher, leaves, fresh[turn & 3], wl.n_label_classes, wl.cls_loss, 5.0, 2.0, 2.0,
                                      class_weight=weight)
        total.backward()
        for o in leaves:
            o["pred_logits"].grad = o["pred_boxes"].grad = None
        return total

    steps = max(10, min(steps, 100))
    for k in range(5):
        step(k)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(steps):
        total = step(k)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    matcher.check_status()
    losses.check_cls_status(device)
    return {"ms_per_step": 1e3 * elapsed / steps, "steps": steps, "loss": float(total.detach()),
            "what": "detr_amd.losses.set_loss_total forward + backward: matching, pair losses, class maps and the "
                    "classification loss of all output sets as one scalar; no synchronisation inside the loop"}


def per_layer_calls(matcher, sets, fresh, device, steps, prefetch=False):
    """The step as the reference's criteria drive it when only the import is swapped: one
    `matcher(output, targets)` call per output set, in a Python loop, every call handing its CPU tuples back
    before the next one starts (/root/reference/models/detr/loss.py:164-169; with a first-stage set
    /root/reference/models/dino_detr/loss.py:165-199).  The targets are new tensor objects at every step and the same
    objects within a step, as in that loop.  A kernel-event pass of its own gives the GPU share.
    `prefetch`: the same loop behind what the forward pre-hook of detr_amd.criterion.attach does when the criterion
    is called -- all sets matched in one batched call, the loop's calls find their results waiting."""
    import detr_amd

    if prefetch:
        def step(turn):
            targets = fresh[turn & 3]
            matcher.prefetch(sets, targets)
            for o in sets:
                indices = matcher(o, targets)       # the loop keeps one set's indices at a time (loss.py:169)
            matcher.clear_prefetched()
            return indices

        for k in range(10):
            step(k)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for k in range(steps):
            held = step(k)
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        del held
        return {"value": len(fresh[0]) * steps / elapsed, "unit": "images/s", "ms_per_step": 1e3 * elapsed / steps,
                "calls_per_step": len(sets),
                "what": "the same loop of per-set calls after detr_amd.criterion.attach(criterion): a forward pre-hook "
                        "matches all sets in one match_many call, the loop's calls are served from it"}

    def step(turn):
        targets = fresh[turn & 3]
        for o in sets:
            indices = matcher(o, targets)           # the loop keeps one set's indices at a time (loss.py:169)
        return indices

    for k in range(10):
        step(k)
    cost_ms, lap_ms = ctypes.c_float(), ctypes.c_float()
    gpu_ms = 0.0
    detr_amd.lib.dm_profile_enable(1)
    try:
        for k in range(max(steps // 4, 10)):
            targets = fresh[k & 3]
            for o in sets:
                matcher(o, targets)
                detr_amd.lib.dm_profile_read(ctypes.byref(cost_ms), ctypes.byref(lap_ms))
                gpu_ms += cost_ms.value + lap_ms.value
    finally:
        detr_amd.lib.dm_profile_enable(0)
    gpu_ms /= max(steps // 4, 10)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(steps):
        held = step(k)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    del held
    return {"value": len(fresh[0]) * steps / elapsed, "unit": "images/s", "ms_per_step": 1e3 * elapsed / steps,
            "calls_per_step": len(sets), "ms_per_call": 1e3 * elapsed / steps / len(sets),
            "kernel_ms_per_step": gpu_ms, "host_and_latency_ms_per_step": 1e3 * elapsed / steps - gpu_ms,
            "what": "for output in sets: matcher(output, targets) -- the reference's criterion loop with only the "
                    "import swapped; every call returns its CPU tuples before the next starts"}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of this
    one (which never touches a GPU), forward rank 0's output, fail if any rank fails."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    children = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                         stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True))
    failed = None
    pending = set(range(n))
    while pending and failed is None:
        for rank in sorted(pending):
            rc = children[rank].poll()
            if rc is None:
                continue
            pending.discard(rank)
            if rc != 0:
                failed = (rank, rc)
                break
        time.sleep(0.05)
    if failed is not None:
        # the other ranks would wait for the dead one in their next fence: end exactly the processes started here
        for rank in pending:
            children[rank].terminate()
        for rank in pending:
            try:
                children[rank].wait(timeout=10)
            except subprocess.TimeoutExpired:
                children[rank].kill()
    out = children[0].stdout.read() if children[0].stdout else ""
    # stdout carries the JSON line only; anything a library of rank 0 printed there (gloo announces its
    # connections on stdout) goes to stderr
    for line in out.splitlines(keepends=True):
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
    sys.stdout.flush()
    if failed is not None:
        raise SystemExit(f"rank {failed[0]} exited with status {failed[1]}")
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--batch", type=int, default=None,
                    help="images per GPU (weak) / in total (strong); default: the workload's")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank matches its own B images; strong: the B images of the workload are "
                         "dealt to the ranks by target count (detr_amd.dist.balanced_shards)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stress", action="store_true", help="skip the cfg4-size cost-kernel measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the passes after the timed regions (flat path, per-layer calls): for profiler runs whose "
                         "per-kernel averages should cover the headline step's launches only")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL); gloo for rehearsals")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: all ranks use cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import detr_amd
    from detr_amd import synth
    from detr_amd.dist import balanced_shards, matched_box_counts, start_num_boxes_all_reduce

    if args.scaling == "strong":
        # the workload's B images in total: every rank draws the same target counts, takes the images
        # balanced_shards deals to it and makes predictions for those only
        import numpy as np

        wl0 = synth.WORKLOADS[args.workload]
        total_images = wl0.batch if args.batch is None else args.batch
        all_counts = synth.target_counts(wl0.t_law, total_images, np.random.default_rng(wl0.seed))
        mine = balanced_shards(all_counts.tolist(), world)[rank]
        wl, sets, targets = synth.make_workload(args.workload, device=device, rank=rank, counts=all_counts[mine])
    else:
        wl, sets, targets = synth.make_workload(args.workload, device=device, batch=args.batch, rank=rank)
        total_images = world * len(targets)
    B = len(targets)
    sizes = [len(t["labels"]) for t in targets]
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    counts = matched_box_counts(sizes, wl.set_queries)

    # Every step is handed target tensors it has not seen in the step before, as a training loop hands over a new
    # batch: the matcher reuses its packing of the previous call only for the very same tensor objects, so a
    # step of the benchmark always packs.  (Views made here, outside the timed region; same storage.)
    fresh = [[{k: v[:] for k, v in t.items()} for t in targets] for _ in range(4)]
    turn = [0]

    # (a gloo rehearsal reduces host tensors: gloo's path for device tensors stages them through pinned memory of its
    # own and stalls for hundreds of milliseconds at random when two ranks share one GPU)
    reduce_on = device if args.backend == "nccl" else "cpu"

    pending = [None]

    def normaliser():
        # the step's one collective, issued while the matching kernels run, on a stream of its own
        pending[0] = start_num_boxes_all_reduce(counts, reduce_on)

    def step():
        turn[0] = (turn[0] + 1) & 3
        indices = matcher.match_many(sets, fresh[turn[0]], while_waiting=normaliser if world > 1 else None)
        if world > 1:
            pending[0].result()     # where the loss needs the value: the stream (not the host) waits for the collective
        return indices

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            if args.backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    # rehearsal hook of the tests: this rank dies here, after its warm-up and before the first fence, where the
    # other ranks will wait for it -- the parent of self-started ranks must end them and exit non-zero
    if os.environ.get("DETR_AMD_BENCH_DIE_RANK") == str(rank):
        os._exit(7)

    def timed_region(steps):
        """EXACTLY `steps` steps between two fences; returns seconds, MAX over ranks."""
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            indices = step()        # held until the next step has returned, as a training loop holds them
        fence()
        seconds = time.perf_counter() - t0
        del indices
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def kernel_times(steps):
        """A pass of its own, outside the headline region: HIP events on the launch stream around the cost
        and the assignment kernels of every step (dm_profile_*), averaged; milliseconds per step."""
        cost_ms, lap_ms = ctypes.c_float(), ctypes.c_float()
        cost_sum = lap_sum = 0.0
        detr_amd.lib.dm_profile_enable(1)
        try:
            for _ in range(steps):
                step()
                # match_many has synchronised the stream: the step's events are complete
                detr_amd.lib.dm_profile_read(ctypes.byref(cost_ms), ctypes.byref(lap_ms))
                cost_sum += cost_ms.value
                lap_sum += lap_ms.value
        finally:
            detr_amd.lib.dm_profile_enable(0)
        return cost_sum / steps, lap_sum / steps

    # Order: the W warm-up steps above, the kernel event-timing pass, then the headline region -- EXACTLY `steps`
    # steps between two fences, Python's cyclic collector ON as in a training loop -- and last the same region with
    # the collector paused (as `timeit` runs; reported beside the headline).  The event-timing pass comes first so that
    # a short run (the driver's --steps 20 --warmup 5) measures the loop in the state a training run is in after its
    # first few milliseconds, not the interpreter's and the pinned allocator's first 25 calls: in a fresh process the
    # host side of a step takes 95 -> 88 us over steps 5..24 and 85 us from step ~40 on, while the kernel takes the
    # same 54 us from step 2 on.  (A gc.collect() right in front of a 20-step region costs it ~10 us per step of cold
    # caches: the paused region runs after the headline for that reason.)  The JSON line carries this as "order".
    roofline = stress = None
    if rank == 0:
        # the two cost-kernel measurements (passes of their own, nothing of them inside the timed region) run FIRST:
        # they are a few milliseconds of sustained GPU work, after which the headline region sees the clocks a
        # training run sees, not those of a process that has just started
        # one kernel per step: the library computes the cost blocks inside the assignment kernel (C <= 4, every
        # T < Q, few enough problems for one workgroup each); asked of the library itself
        from detr_amd._lib import DmSet
        arr = (DmSet * len(sets))()
        for s_, q in enumerate(wl.set_queries):
            arr[s_].Q, arr[s_].q_stride = q, (q + 31) & ~31
        cost_sets = ctypes.c_int(-1)
        rc = detr_amd.lib.dm_match_plan(arr, len(sets), B, wl.n_logits, max(sizes), ctypes.byref(cost_sets))
        assert rc == 0, rc
        fused = cost_sets.value == 0
        # logits + boxes of every prediction, labels + boxes of every target per set, two int64 indices per match
        step_bytes = sum(B * q * (wl.n_logits * 4 + 16) + sum(sizes) * 24 + 16 * sum(min(q, t) for t in sizes)
                         for q in wl.set_queries)
        # The cost kernel on this run's inputs.  When the step runs as ONE kernel (C <= 4: every assignment
        # workgroup computes its own cost block in LDS, no cost matrix in HBM) the cost kernel is not part of
        # the headline step any more; it is still the kernel the cost-kernel HBM figure of the metric is
        # about, and the one that runs for C > 4, T >= Q and return_cost=True.
        roofline = stress_roofline(device, launches=max(args.steps, 50), copy_reference=False,
                                   given=(wl, sets, targets))
        roofline["part_of_headline_step"] = not fused
        if world == 1 and not args.no_stress:
            stress = stress_roofline(device)
    cost_ms_avg, lap_ms_avg = kernel_times(max(args.steps, 50))
    # One short untimed region bracketed exactly like the timed one.  The first steps after a device-wide
    # synchronisation that follows a long unsynchronised run take 90-115 us instead of 73-76 (per-step stamps,
    # tools/warmcurve.py: steps 2-7 behind the fence, whatever ran before it); a timed region of 20 steps would carry
    # that one-off as +8 us per step (83-86 us against 75-77 us for every later 20-step region of the same process).
    timed_region(min(5, args.steps))
    elapsed = timed_region(args.steps)
    gc.collect()
    gc.disable()
    try:
        timed_region(min(5, args.steps))      # (a full collection empties the interpreter's free lists: refilled here, untimed)
        elapsed_paused = timed_region(args.steps)
    finally:
        gc.enable()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        record = {
            "metric": "matched images/sec (Q=600, ~16 GT/img, 6 aux layers); cost-kernel HBM GB/s",
            "value": total_images * args.steps / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{wl.name}: B={B} images per GPU ({total_images} in total), sets Q={wl.set_queries}, C={wl.n_logits}, "
                                   f"{wl.cls_loss} class cost, T~{wl.t_law} (sum T={sum(sizes)}, max T={max(sizes)}), "
                                   f"{len(wl.set_queries) * B} assignment problems per step per GPU",
                       "parallelism": f"dp{world}", "python_gc": "on",
                       "order": f"{args.warmup} warm-up steps, the cost-kernel passes (roofline, roofline_stress), "
                                f"kernel-event pass ({max(args.steps, 50)} steps), a fenced rehearsal region ({min(5, args.steps)} steps, untimed), "
                                f"the headline region ({args.steps} steps), the collector-paused region ({args.steps} steps)"},
            "python_gc_paused": {"value": total_images * args.steps / elapsed_paused, "unit": "images/s",
                                 "ms_per_step": 1e3 * elapsed_paused / args.steps},
            "roofline": roofline,
            "step_kernels": {"what": "HIP events on the launch stream around the kernels of every step, in a pass of "
                                     "its own (dm_profile_*)",
                             "one_kernel_per_step": fused,
                             # (with one kernel per step there is no cost kernel between the first two events)
                             "cost_kernel_ms_per_step": None if fused else cost_ms_avg,
                             "assignment_kernel": ("lap_block_kernel<FUSED>: cost block computed in LDS + assignment"
                                                   if fused else "lap_block_kernel / lap_wave_kernel"),
                             "assignment_kernel_ms_per_step": lap_ms_avg,
                             "assignment_kernel_traffic_bytes": (measured_traffic("lap_block_kernel", wl.name)
                                                                 if fused and B == wl.batch else None),
                             "problems_per_s": len(wl.set_queries) * B / (lap_ms_avg * 1e-3),
                             # what this kernel has to move when it computes its own cost blocks: predictions and
                             # targets in, indices out -- a latency-bound kernel, its time is the dependent chain
                             # of the heaviest problem of the step, not its traffic
                             "algorithmic_bytes": step_bytes if fused else None,
                             "achieved_gbs": (step_bytes / (lap_ms_avg * 1e-3) / 1e9) if fused and lap_ms_avg else None,
                             "frac_of_hbm_peak": (step_bytes / (lap_ms_avg * 1e-3) / 1e9 / 8000.0) if fused and lap_ms_avg else None,
                             "bound": "latency (one workgroup per problem; the step takes as long as its heaviest problem)"},
        }
        if world == 1 and not args.no_extras:
            record["flat_path"] = flat_path(matcher, sets, targets, device, args.steps)
            record["per_layer_calls"] = per_layer_calls(matcher, sets, fresh, device, args.steps)
            record["per_layer_calls_prefetched"] = per_layer_calls(matcher, sets, fresh, device, args.steps, prefetch=True)
            record["criterion_step"] = criterion_step(matcher, wl, sets, fresh, device, args.steps)
        if stress is not None:
            record["roofline_stress"] = stress
        if world == 1 and not args.no_cpu_baseline:
            record["cpu_baseline"] = cpu_baseline(wl, sets, targets)
        print(json.dumps(record), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
