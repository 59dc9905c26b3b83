#!/usr/bin/env python3
"""bench.py -- Mblocks/s of the hot path (MF::calcMotionBlockMatching) on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one whole pyramid (search + regulariser of every level + dense expansion) over one
synthetic 4K frame pair per GPU (BASELINE.json configs[2]; configs[4] for N > 1: one pair per
GPU, dense .flo fields gathered to rank 0 over RCCL).  Frames are resident in HBM before the
timed region; the pyramid build (the reference's constructor, outside its own timed region,
main_class.cpp:45-55) is not part of a step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto 4 hardware queues by default; the `sequence` leg keeps more pairs than that in flight
# (one stream each).  Must be set before the HIP runtime starts; no effect on the single-pair `value`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_SAD_PEAK_T = 157.3               # T abs-diff/s: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz x 4 (v_sad_u8), SURVEY.md 8d

WORKLOADS = {
    # name: (width, height, search_size, block_size, levels, description)
    "cfg3": (3840, 2160, 80, 16, 4, "cfg3: 4K 3840x2160 synthetic pair, 16x16 blocks, +-32 spiral full search, "
                                     "4-level pyramid, 8-neighbour regulariser (2 sweeps per block size 16..2)"),
    "cfg2": (1920, 1080, 48, 16, 3, "cfg2: 1080p synthetic pair, 16x16 blocks, +-16, 3 levels"),
    "cfg4": (3840, 2160, 72, 8, 4, "cfg4: 4K synthetic pair, 8x8 blocks, +-32, 4 levels"),
    "cfg1": (584, 388, 30, 16, 3, "cfg1: RubberWhale-sized synthetic pair, 16x16 blocks, +-7, 3 levels"),
    "ref": (2336, 1552, 64, 32, 4, "reference literals (main_class.cpp:19-21): 584x388 frame up-sampled x4, 32x32 blocks, "
                                   "search 64 (+-16), 4 levels"),
    # the author's OTHER literal set, commented out at main_class.cpp:15-17 (index 0 = finest level): mixed block sizes, and an
    # odd shift at the coarsest level (42 - 32 = 10 -> R = 5) with 32 x 32 blocks over 16 x 16 ones
    "ref2": (584, 388, [32, 32, 42], [16, 16, 32], 3, "reference's second literal set (main_class.cpp:15-17): 584x388 frame, block "
                                                       "{16,16,32}, search {32,32,42} (+-8, +-8, +-5), 3 levels"),
}
HEADLINE_WORKLOADS = sorted(k for k, v in WORKLOADS.items() if not isinstance(v[2], list))   # --workload: one block size / range
# pairs in flight of the `sequence_deep` legs (4 batched contexts): small frames are launch-bound, so they take deeper batches
DEEP_SEQUENCE = {"cfg1": 64, "cfg2": 32, "cfg4": 24, "ref": 32, "ref2": 64}


def per_level(v, levels):
    """search_size[] / block_size[] of a workload (index 0 = finest level) from one value or a list"""
    return list(v)[:levels] if isinstance(v, (list, tuple)) else [v] * levels


def level_blocks(pw, ph, block, levels):
    return [((pw >> l) // b) * ((ph >> l) // b) for l, b in enumerate(per_level(block, levels))]


def check_all_pairs(bbme, ctxs, frames, per, search, block, levels, device):
    """Every pair of every batched context against a context of its own on the same frames (the int16 2x2-cell grids: exactly
    the information of the dense field).  Outside every timed region."""
    mf, ok = None, True
    for i, (a1, a2) in enumerate(frames):
        if mf is None:
            mf = bbme.MF(a1, a2, per_level(search, levels), per_level(block, levels), levels, device=device, frames_on_device=True)
            # no second (lowest-priority) stream: a context that speculates leaves the following legs' streams on slower hardware
            # queues (measured: the sequence legs behind a speculating checker lose 10-65 %)
            mf.set_speculation(False)
        else:
            mf.set_frames_device(a1, a2)
        mf.estimate_async()
        mf.synchronize()
        ok = ok and bool(np.array_equal(mf.get_cells(), ctxs[i // per].get_pair_cells(i % per)))
    if mf is not None:
        mf.close()
    return ok


def pmc_traffic(levels):
    """(mean HBM-side bytes per search launch, where the figure comes from) from the committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_search.json, made by scripts/pmc_report.py: FETCH_SIZE scaled by the factor
    measured on a calibration read of known size in the same run, plus WRITE_SIZE).  It is a QUOTE of a committed
    measurement, not a measurement of this run: only given when the file was measured on exactly this kernel
    source (sha256 of bbme_kernels.hpp); otherwise (None, None)."""
    import glob
    import hashlib
    src = os.path.join(ROOT, "blockbasedmotionestimation_amd", "csrc", "bbme_kernels.hpp")
    digest = hashlib.sha256(open(src, "rb").read()).hexdigest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_search.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("kernel_source_sha256") != digest:
                continue
            vals = [v["hbm_bytes"] for k, v in d["kernels"].items() if k.startswith("k_search")]
            if len(vals) == levels:
                return sum(vals) / len(vals), {"file": os.path.relpath(path, ROOT), "kernel_source_sha256": digest[:16],
                                               "kind": "quoted from a committed rocprofv3 --pmc pass on this kernel source, "
                                                       "not measured in this run"}
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def pmc_sq_summary():
    """The SQ-counter summary of the level-0 search launch from the committed rocprofv3 --pmc passes (profiles/rNN_pmc_search_sq.json,
    scripts/pmc_search.sh + pmc_search_report.py): VALU busy, instruction mix, LDS conflicts, waits.  Only quoted when measured on
    exactly this kernel source."""
    import glob
    import hashlib
    src = os.path.join(ROOT, "blockbasedmotionestimation_amd", "csrc", "bbme_kernels.hpp")
    digest = hashlib.sha256(open(src, "rb").read()).hexdigest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_search_sq.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("kernel_source_sha256") == digest and "summary_level0_launch" in d:
                out = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["summary_level0_launch"].items()}
                out["source"] = os.path.relpath(path, ROOT)
                return out
        except (OSError, ValueError, KeyError):
            pass
    return None


def epe_on_ground_truth(bbme, device, jacobi=False):
    """Average end-point error of the reference's own pipeline and literals (4x bilinear up-sampling, 4
    levels, block 32, search 64, every 4th pixel / 4; main_class.cpp:19-21,32-33,58-70) against Middlebury
    ground truth.  The Middlebury frames are not in the reference (only its GT .flo files are), so the pair
    is a texture warped by the Venus ground truth, which makes that file the true flow of the pair."""
    gt_path = os.path.join(ROOT, "tests", "golden", "gt_Venus_flow10.flo")
    if not os.path.exists(gt_path):
        return None
    gt = bbme.Flow().ReadFlowFile(gt_path)
    h, w = gt.shape[:2]
    f1, f2 = bbme.warp_pair_from_flow(gt)
    u1, u2 = bbme.resize_x4(f1), bbme.resize_x4(f2)
    mf = bbme.MF(u1, u2, [64] * 4, [32] * 4, 4, device=device)
    mf.set_regularizer_mode(jacobi)
    flow = mf.calcMotionBlockMatching()
    sub = bbme.subsample_div4(flow, mf.padding_x, mf.padding_y, w, h)
    mf.close()
    return {"value": round(bbme.Flow().CalculateMSE(gt, sub), 6), "unit": "px",
            "data": "texture warped by Middlebury Venus flow10.flo (420x380); reference pipeline: x4 bilinear, "
                    "4 levels, 32x32 blocks, search 64"}


def cgroup_cpu_quota():
    """(cores, text): the CPU-time quota of this job's control group in cores (None = unlimited) and the raw setting read --
    cgroup v2 cpu.max ("max 100000" / "1600000 100000"), else v1 cpu.cfs_quota_us / cpu.cfs_period_us."""
    try:
        text = open("/sys/fs/cgroup/cpu.max").read().strip()
        quota, period = text.split()[:2]
        return (None if quota == "max" else float(quota) / float(period)), "cpu.max: " + text
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return (None if quota <= 0 else quota / period), "cpu.cfs_quota_us / cpu.cfs_period_us: %d / %d" % (quota, period)
    except (OSError, ValueError):
        return None, "no cgroup cpu quota file"


def cpu_info():
    """model name, nproc, cores this job may really use = min(affinity mask, cgroup quota), and the quota as read."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    quota, quota_text = cgroup_cpu_quota()
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.5)))
    return model, os.cpu_count(), usable, quota_text


_NATIVE_ORACLE = {}


def native_oracle(omp):
    """The oracle's C restatement rebuilt for this host (-O3 -march=native; with OpenMP over the macroblocks of the search
    when `omp`), loaded in place of the portable build.  Checker / CPU baseline only."""
    from oracle import bbme_oracle as O
    if omp not in _NATIVE_ORACLE:
        src = os.path.join(ROOT, "oracle", "bbme_oracle.c")
        out = os.path.join(tempfile.mkdtemp(prefix="bbme_cpu_"), "liboracle_native_%s.so" % ("omp" if omp else "st"))
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-ffp-contract=off"] +
                              (["-fopenmp"] if omp else []) + ["-o", out, src, "-lm"])
        _NATIVE_ORACLE[omp] = out
    O._LIB_PATH = _NATIVE_ORACLE[omp]
    O._lib = None
    return O


def oracle_flow(f1, f2, search, block, levels, threads):
    """(seconds, field) of the whole pyramid on the CPU oracle, timed as the reference times calcMotionBlockMatching
    (main_class.cpp:47-55): threads == 1 like the reference; threads > 1: the search's macroblock loop spread with OpenMP
    (the regulariser sweeps stay sequential: they are order dependent)."""
    O = native_oracle(threads > 1)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    omf = O.OracleMF(f1, f2, per_level(search, levels), per_level(block, levels), use_cache=False)
    t0 = time.perf_counter()
    flow = omf.calc_motion_block_matching()
    dt = time.perf_counter() - t0
    omf.close()
    return dt, flow


def cpu_pool():
    # every host core this job may use (affinity mask and cgroup quota); BBME_CPU_THREADS overrides
    return max(1, int(os.environ.get("BBME_CPU_THREADS", cpu_info()[2])))


def cpu_baseline(f1, f2, search, block, levels, expect_flow):
    """The oracle timed on the same pair, whole pyramid: (i) one thread, like the reference; (ii) all usable host cores.
    Returns [(seconds, threads, parity)] for the two legs."""
    legs = []
    for threads in (1, cpu_pool()):
        dt, flow = oracle_flow(f1, f2, search, block, levels, threads)
        parity = bool(np.array_equal(flow, expect_flow)) if expect_flow is not None else None
        legs.append((dt, threads, parity))
    return legs


def other_workload(bbme, torch, name, device, steps, warmup, check):
    """One of the other single-GPU BASELINE configs, timed as `value` is (frames resident, graph replays, device sync on both
    sides), a few steps; search / regulariser split from one eager HIP-event pass; `parity_vs_oracle` = the field against
    the CPU oracle's on the same pair (all host cores), None when the CPU legs are switched off."""
    w, h, search, block, levels, desc = WORKLOADS[name]
    f1, f2, _ = bbme.synth_pair(w, h, 1000 + 30, max_motion=24)
    mf = bbme.MF(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda(), per_level(search, levels), per_level(block, levels), levels,
                 device=device, frames_on_device=True)
    mf.synchronize()
    for _ in range(warmup):
        mf.estimate_async()
    mf.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        mf.estimate_async()
    mf.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flow = mf.get_flow()
    mf.set_profiling(True)
    mf.estimate_async()
    prof = mf.timings()
    mf.set_profiling(False)
    blocks = level_blocks(mf.padded_width, mf.padded_height, block, levels)
    mf.close()
    parity = None
    if check:
        parity = bool(np.array_equal(oracle_flow(f1, f2, search, block, levels, cpu_pool())[1], flow))
    out = {"workload": desc, "value": round(blocks[0] / dt / 1e6, 4), "unit": "Mblocks/s", "ms_per_step": round(dt * 1e3, 4),
           "steps": steps, "blocks_level0": blocks[0], "search_ms": round(prof["search_ms"], 4),
           "regularize_ms": round(prof["regularize_ms"], 4), "parity_vs_oracle": parity}
    # the same workload as a sequence: 8 pairs in flight as 4 batched contexts of 2 (what `sequence` is for the headline workload).
    # Small frames are launch-bound one pair at a time; side by side the launches are shared.
    frames = [(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda())]
    for k in range(1, 8):
        g1, g2, _ = bbme.synth_pair(w, h, 1000 + 30 + k, max_motion=24)
        frames.append((torch.from_numpy(g1).cuda(), torch.from_numpy(g2).cuda()))
    ctxs = [bbme.MFBatch(frames[2 * i:2 * i + 2], per_level(search, levels), per_level(block, levels), levels, device=device, frames_on_device=True)
            for i in range(4)]
    for c in ctxs:
        c.set_speculation(False)
        c.estimate_async()
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for c in ctxs:
            c.estimate_async()
    for c in ctxs:
        c.synchronize()
    dts = (time.perf_counter() - t0) / (steps * 8)
    same = bool(np.array_equal(ctxs[0].get_pair_flow(0), flow))
    all_ok = check_all_pairs(bbme, ctxs, frames[:8], 2, search, block, levels, device)
    for c in ctxs:
        c.close()
    out["sequence_8_pairs"] = {"value": round(blocks[0] / dts / 1e6, 4), "unit": "Mblocks/s", "ms_per_pair": round(dts * 1e3, 4),
                               "contexts": 4, "pairs_per_context": 2, "first_pair_field_unchanged": same,
                               "all_pairs_checked": all_ok}
    # a longer sequence on the same four streams: more pairs per batched context (the same eight pairs rolled by a few pixels)
    deep = DEEP_SEQUENCE.get(name, 0)
    if deep > 8:
        per = deep // 4
        while len(frames) < deep:
            k = len(frames)
            sh = (3 * (k // 8), 5 * (k // 8))
            frames.append((torch.roll(frames[k % 8][0], sh, (0, 1)).contiguous(), torch.roll(frames[k % 8][1], sh, (0, 1)).contiguous()))
        ctxs = [bbme.MFBatch(frames[per * i:per * (i + 1)], per_level(search, levels), per_level(block, levels), levels, device=device,
                             frames_on_device=True) for i in range(4)]
        for c in ctxs:
            c.set_speculation(False)
            c.estimate_async()
        for c in ctxs:
            c.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for c in ctxs:
                c.estimate_async()
        for c in ctxs:
            c.synchronize()
        dts = (time.perf_counter() - t0) / (steps * deep)
        same = bool(np.array_equal(ctxs[0].get_pair_flow(0), flow))
        all_ok = check_all_pairs(bbme, ctxs, frames[:4 * per], per, search, block, levels, device)
        for c in ctxs:
            c.close()
        out["sequence_deep"] = {"pairs_in_flight": deep, "value": round(blocks[0] / dts / 1e6, 4), "unit": "Mblocks/s",
                                "ms_per_pair": round(dts * 1e3, 4), "contexts": 4, "pairs_per_context": per,
                                "first_pair_field_unchanged": same, "all_pairs_checked": all_ok,
                                "note": "pairs beyond the 8 synthesised ones are the same pairs rolled by a few pixels"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=HEADLINE_WORKLOADS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-iters", type=int, default=5)
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the `other_workloads` legs (cfg2, cfg4, cfg1, ref, ref2)")
    ap.add_argument("--no-host-boundary", action="store_true",
                    help="skip the host_boundary legs (for kernel traces: their pyramids start from freshly uploaded frames)")
    ap.add_argument("--in-flight", type=int, default=8,
                    help="N=1 only: also report the throughput of a sequence with this many independent pairs in flight "
                         "(batched contexts of --seq-batch pairs); 0 = skip.  Reported beside `value`, never as `value`")
    ap.add_argument("--seq-batch", type=int, default=2, help="pairs per batched context in the `sequence` leg")
    ap.add_argument("--in-flight-deep", type=int, default=24,
                    help="N=1 only: a second sequence leg with this many pairs in flight, as 4 batched contexts (`sequence_deep`); "
                         "0 = skip")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on one GPU: run the N>1 code path (process group, gather, expand) with world size 1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    import blockbasedmotionestimation_amd as bbme

    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when a communicator is created; the contract is ONE JSON line there
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    w, h, search, block, levels, desc = WORKLOADS[args.workload]
    f1, f2, _ = bbme.synth_pair(w, h, 1000 + 30 + rank, max_motion=24)      # one pair per GPU (weak scaling)
    t1 = torch.from_numpy(f1).cuda()
    t2 = torch.from_numpy(f2).cuda()
    mf = bbme.MF(t1, t2, per_level(search, levels), per_level(block, levels), levels, device=local_rank, frames_on_device=True)
    mf.synchronize()
    pw, ph = mf.padded_width, mf.padded_height
    blocks = level_blocks(pw, ph, block, levels)

    # Multi-GPU (sequence.CellGather): each rank's result travels as the compact cell grid -- one int16 (dx, dy) pair
    # per 2x2 cell, 4.2 MB at 4K, exactly the information of the dense field -- in one gather to rank 0, which expands
    # every gathered grid to the dense .flo field (copy_to_all_pixels) on a second stream beside the next estimate.
    gather = None
    if use_dist:
        from blockbasedmotionestimation_amd.sequence import mf_cell_gather
        # an explicit stream: torch's default stream has handle 0, which bbme_set_stream reads as "create a private stream"
        work_stream = torch.cuda.Stream(device=local_rank)
        torch.cuda.set_stream(work_stream)
        # Two pairings of the context's graph with the gather (sequence.CellGather): the speculative graph with gather and
        # expansions in order on its stream, or the plain graph with gather and expansions on a second stream beside the
        # next estimate -- a second stream waiting behind the forked (speculative) graph costs more than it hides
        # (scripts/dist_step_probe.py).  Both give the same fields; which is faster depends on N (the gather and rank 0's N
        # expansions grow with it), so both are timed here, before the timed region, and every rank keeps the faster.
        dist_modes = [("speculative graph; gather and expansions in order on its stream", True, False),
                      ("plain graph; gather and expansions on a second stream beside the next estimate", False, True)]
        dist_calibration = []
        for text, spec, overlap in dist_modes:
            mf.set_speculation(spec)
            cand = mf_cell_gather(mf, local_rank, overlap=overlap)
            for _ in range(3):
                cand.step()
            cand.fence(); mf.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                cand.step()
            cand.fence(); mf.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / 10 * 1e3], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist_calibration.append((float(t.item()), text, spec, overlap))
            del cand
        _, dist_mode_text, spec, overlap = min(dist_calibration)        # all-reduced times: the same choice on every rank
        mf.set_speculation(spec)
        gather = mf_cell_gather(mf, local_rank, overlap=overlap)

    def step():
        if gather is None:
            mf.estimate_async()                   # whole pyramid, no host wait
        else:
            gather.step()

    def fence():
        if gather is not None:
            gather.fence()
        mf.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    result_flow = mf.get_flow() if rank == 0 else None
    if use_dist and rank == 0:
        # the field rank 0 expanded from its own gathered cells must be the field its context holds
        assert np.array_equal(gather.flows[0].cpu().numpy(), result_flow), "gathered + expanded field differs"

    # a sequence on one GPU: K independent pairs in flight, one context (and private stream) each.
    # One pair leaves most of the chip idle while the regulariser walks its dependency chains, so
    # pairs of a sequence overlap well.  Reported as `sequence`, beside the single-pair `value`.
    sequence = sequence_deep = None
    if rank == 0 and not use_dist and args.in_flight > 1:
        # Pairs that share a GPU go into BATCHED contexts (bbme_create_batch: every kernel works on all pairs of the context
        # at once): the device dispatches the dependent kernels of many streams no faster than one per ~4 us chip-wide, so
        # one context per pair is dispatch-bound (69 launches per pair); `--seq-batch` pairs per context, in_flight / batch
        # contexts (streams) side by side so that one context's searches overlap the others' latency-bound sweeps.
        seq_frames = [(t1, t2)]
        for k in range(1, args.in_flight):
            g1, g2, _ = bbme.synth_pair(w, h, 1000 + 30 + k, max_motion=24)
            seq_frames.append((torch.from_numpy(g1).cuda(), torch.from_numpy(g2).cuda()))

        def sequence_leg(n_in_flight, per):
            per = max(1, min(per, n_in_flight))
            n_ctx = max(1, n_in_flight // per)
            frames = list(seq_frames[:n_ctx * per])
            while len(frames) < n_ctx * per:      # deeper than the synthesised set: the same pairs rolled by a few pixels (new content,
                k = len(frames)                   # same statistics; synthesising a 4K pair takes seconds of CPU time)
                a1, a2 = seq_frames[k % len(seq_frames)]
                sh = (3 * (k // len(seq_frames)), 5 * (k // len(seq_frames)))
                frames.append((torch.roll(a1, sh, (0, 1)).contiguous(), torch.roll(a2, sh, (0, 1)).contiguous()))
            ctxs = [bbme.MFBatch(frames[i * per:(i + 1) * per], per_level(search, levels), per_level(block, levels), levels, device=local_rank,
                                 frames_on_device=True) for i in range(n_ctx)]
            for c in ctxs:
                c.set_speculation(False)              # with pairs in flight the chip is busy anyway
                c.estimate_async()
            for c in ctxs:
                c.synchronize()
            seq_steps = max(2, args.steps // 2)
            t0 = time.perf_counter()
            for _ in range(seq_steps):
                for c in ctxs:
                    c.estimate_async()
            for c in ctxs:
                c.synchronize()
            dt = time.perf_counter() - t0
            same = bool(np.array_equal(ctxs[0].get_pair_flow(0), result_flow))
            all_ok = check_all_pairs(bbme, ctxs, frames, per, search, block, levels, local_rank)
            n_pairs = n_ctx * per
            out = {"pairs_in_flight": n_pairs, "contexts": n_ctx, "pairs_per_context": per,
                   "value": round(blocks[0] * n_pairs * seq_steps / dt / 1e6, 4),
                   "unit": "Mblocks/s", "ms_per_pair": round(dt / (seq_steps * n_pairs) * 1e3, 4),
                   "pairs": seq_steps * n_pairs, "first_pair_field_unchanged": same, "all_pairs_checked": all_ok,
                   "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}
            for c in ctxs:
                c.close()
            return out

        sequence = sequence_leg(args.in_flight, args.seq_batch)
        if args.in_flight_deep > args.in_flight:
            # a longer sequence: more pairs per context on the same four streams (each launch carries more work)
            sequence_deep = sequence_leg(args.in_flight_deep, max(1, args.in_flight_deep // 4))
            sequence_deep["note"] = ("pairs beyond the %d synthesised ones are the same pairs rolled by a few pixels (new content, same "
                                     "statistics)" % args.in_flight)
        del seq_frames

    # the boundary with host buffers (never `value`): frames in (pinned) host memory -> upload (2 x 8.3 MB at 4K), padding +
    # pyramid on the GPU, estimate, download of the dense field (66.8 MB) or of the compact 2x2-cell grid (4.2 MB)
    host_boundary = None
    if rank == 0 and not use_dist and not args.no_host_boundary:
        p1, p2 = torch.from_numpy(f1).pin_memory(), torch.from_numpy(f2).pin_memory()
        reps = 9

        def through_host(get, reps=reps, drain=None):
            get(0); get(1)                        # first touch of the pinned buffers, files created
            if drain:
                drain()
            mf.synchronize()
            times = []
            t_all = time.perf_counter()
            for i in range(reps):
                t0 = time.perf_counter()
                t1.copy_(p1, non_blocking=True)
                t2.copy_(p2, non_blocking=True)
                mf.set_frames_device(t1, t2)      # orders the context's stream behind the two uploads
                mf.estimate_async()
                get(i)
                times.append(time.perf_counter() - t0)
            if drain:                             # pipelined legs: whole wall time, the last files included
                drain()
                return (time.perf_counter() - t_all) / reps * 1e3
            return float(np.median(times)) * 1e3  # median: the PCIe link of a shared host is noisy
        pin_flow = [torch.empty((ph, pw, 2), dtype=torch.float32).pin_memory().numpy() for _ in range(2)]
        pin_cells = torch.empty((ph // 2, pw // 2, 2), dtype=torch.int16).pin_memory().numpy()
        dense_ms = through_host(lambda i: mf.get_flow(pin_flow[0]))
        cells_ms = through_host(lambda i: mf.get_cells(pin_cells))
        # end to end as the reference's driver would be with its writer called (main_class.cpp:24-75 + rw_flow.cpp:139-200):
        # ... download, strip the padding, Flow::WriteFlowFile -- synchronously, and with the asynchronous writer
        # (bbme_flo_writer_*: the file of pair i is written from pinned memory while pair i + 1 is estimated)
        out_dir = tempfile.mkdtemp(prefix="bbme_flo_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        flo = bbme.Flow()

        def write_sync(i):
            fl = mf.get_flow(pin_flow[0])
            flo.WriteFlowFile(fl[mf.padding_y:mf.padding_y + h, mf.padding_x:mf.padding_x + w], os.path.join(out_dir, "s%d.flo" % i))
        sync_ms = through_host(write_sync)
        writer = bbme.FlowWriter()

        def write_async(i):
            writer.wait()                             # the previous pair's file is complete (written during this estimate)
            fl = mf.get_flow(pin_flow[i & 1])
            writer.submit(fl, os.path.join(out_dir, "a%d.flo" % i), mf.padding_x, mf.padding_y, w, h)
        async_ms = through_host(write_async, reps=8, drain=writer.wait)
        # ... and from the compact result: 4.2 MB of cells cross PCIe, the worker expands + strips + writes (submit_cells)
        # with `depth` writers (one file each in flight: separate files do not share an inode lock)
        depth = int(os.environ.get("BBME_BENCH_WRITERS", "3"))
        pin_cells2 = [pin_cells] + [torch.empty((ph // 2, pw // 2, 2), dtype=torch.int16).pin_memory().numpy() for _ in range(depth - 1)]
        writers = [writer] + [bbme.FlowWriter() for _ in range(depth - 1)]

        def write_cells_async(i):
            writers[i % depth].wait()                 # buffer i % depth is free again
            cl = mf.get_cells(pin_cells2[i % depth])
            writers[i % depth].submit_cells(cl, os.path.join(out_dir, "c%d.flo" % i), mf.padding_x, mf.padding_y, w, h)
        cells_async_ms = through_host(write_cells_async, reps=24, drain=lambda: [wr.wait() for wr in writers])
        for wr in writers[1:]:
            wr.close()
        t0 = time.perf_counter()
        writer.submit_cells(pin_cells2[0], os.path.join(out_dir, "alone.flo"), mf.padding_x, mf.padding_y, w, h)
        writer.wait()
        cells_write_ms = (time.perf_counter() - t0) * 1e3
        same_file = same_file_c = open(os.path.join(out_dir, "s0.flo"), "rb").read() == open(os.path.join(out_dir, "c0.flo"), "rb").read()
        same_file = same_file_c and open(os.path.join(out_dir, "s0.flo"), "rb").read() == open(os.path.join(out_dir, "a0.flo"), "rb").read()
        writer.close()
        import shutil
        shutil.rmtree(out_dir, ignore_errors=True)
        host_boundary = {"host_frames_to_dense_field_ms": round(dense_ms, 2), "host_frames_to_cells_ms": round(cells_ms, 2),
                         "value_with_dense_download": round(blocks[0] / dense_ms / 1e3, 3),
                         "value_with_cells_download": round(blocks[0] / cells_ms / 1e3, 3), "unit": "Mblocks/s",
                         "end_to_end_ms": round(sync_ms, 2), "end_to_end_async_writer_ms": round(async_ms, 2),
                         "end_to_end_cells_writer_ms": round(cells_async_ms, 2), "cells_writer_alone_ms": round(cells_write_ms, 2),
                         "writers_in_flight": depth,
                         "value_end_to_end": round(blocks[0] / min(async_ms, cells_async_ms) / 1e3, 3),
                         "flo_bytes": 12 + 8 * w * h, "flo_files_identical": same_file,
                         "note": "per pair: upload of the two frames (pinned), padding + pyramid on the GPU, estimate, download "
                                 "(pinned); end_to_end adds stripping the padding and Flow::WriteFlowFile to a tmpfs file, "
                                 "synchronously / on the writer thread beside the next pair / on the writer thread from the compact cell grid (4.2 MB "
                                 "over PCIe, expansion fused into the write, writers_in_flight files at a time); value_end_to_end is the best of them"}

    # per-kernel timing with HIP events on the ctx stream (eager launches, same kernels and data)
    prof = None
    if rank == 0:
        mf.set_profiling(True)
        acc = {}
        for _ in range(args.profile_iters):
            mf.estimate_async()
            for k, val in mf.timings().items():
                acc[k] = acc.get(k, 0.0) + val
        mf.set_profiling(False)
        prof = {k: val / args.profile_iters for k, val in acc.items()}

    if rank == 0:
        R = (search - block) >> 1
        units = blocks[0] * world
        # the binding ceiling is the issue rate of the SAD instructions; measure it on this device
        import ctypes as C
        from blockbasedmotionestimation_amd import _capi
        rates = (C.c_double * 4)()
        _capi.check(_capi.lib().bbme_probe_rates(local_rank, rates))
        qsad_peak = rates[0] * 64 * 16 / 1e3          # T abs-diff/s through v_qsad_pk_u16_u8, all operands in VGPRs
        sad_peak = rates[1] * 64 * 4 / 1e3            # T abs-diff/s through v_sad_u8
        loops = (C.c_double * 2)()
        _capi.check(_capi.lib().bbme_probe_search_loops(local_rank, loops))
        value = units * args.steps / elapsed / 1e6
        # dominant kernel (by arithmetic): k_search_fast<B, W>, one launch per level.  Algorithmic bytes per block
        # = B^2 + (B+2R)^2 + 8; "per launch" = mean over the `levels` launches of a pyramid.
        bytes_per_block = block * block + (block + 2 * R) ** 2 + 8
        search_bytes = sum(blocks) * bytes_per_block / levels
        search_ms = prof["search_ms"] / levels
        achieved = search_bytes / (search_ms * 1e-3) / 1e9
        absdiff = sum(blocks) * (2 * R + 1) ** 2 * block * block / levels
        tabs = absdiff / (search_ms * 1e-3) / 1e12
        seen, viol = C.c_int(), C.c_int()
        _capi.check(_capi.lib().bbme_probe_xcd(local_rank, C.byref(seen), C.byref(viol)))
        xcd_check = {"xcds_seen": seen.value, "workgroups_off_their_residue_class": viol.value, "of": 4096}
        traffic, traffic_source = pmc_traffic(levels)
        ms_step = elapsed / args.steps * 1e3
        step_absdiff = sum(blocks) * (2 * R + 1) ** 2 * block * block * world      # the searches of one step, all levels
        step_tabs = step_absdiff / (ms_step * 1e-3) / 1e12
        out = {
            "metric": "Mblocks/s (16x16, +-32 full search)", "value": round(value, 4), "unit": "Mblocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc, "pairs_per_step": world, "frame": [w, h], "padded": [pw, ph],
                       "block": block, "search_range": R, "levels": levels,
                       "blocks_level0": blocks[0], "blocks_all_levels": sum(blocks),
                       "multi_gpu": ("one pair per GPU; int16 cell grids gathered on rank 0 over RCCL and expanded there to "
                                     "the dense .flo fields (" + dist_mode_text + "; ms per step measured before the timed "
                                     "region: " + ", ".join("%.3f %s" % (t, "speculative" if sp else "plain/overlapped")
                                                            for t, _, sp, _ in dist_calibration) + ")")
                                    if use_dist else "single GPU"},
            # contract fields (achieved / peak / unit / frac / traffic) are the HBM figures; what BINDS the kernel is the issue
            # rate of the integer SAD instructions, priced in `binding` against the chip's spec rate
            "roofline": {"bound": "valu", "kernel": "k_search_fast<%d, W> (mean of the %d per-level launches; W = 2 waves per macroblock on levels of "
                                   "<= 10000 blocks, else 1)" % (block, levels),
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": round(search_bytes),
                         "avg_launch_ms": round(search_ms, 5),
                         "binding": {"unit": "T abs-diff/s", "achieved": round(tabs, 3), "peak": VALU_SAD_PEAK_T,
                                     "frac": round(tabs / VALU_SAD_PEAK_T, 5),
                                     "peak_source": "256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz x 4 abs-diff per v_sad_u8 lane-op: the "
                                                    "ceiling of the SAD INSTRUCTIONS, not of the VALU -- v_sad_u8 holds its SIMD for 4 cycles "
                                                    "per wave64 instruction (v_qsad_pk_u16_u8 for 16) although the SIMD retires v_fma_f32 "
                                                    "at 32 lanes/clk; the v_sad_u8 probe below (137-140 T/s at the 2.1-2.2 GHz the chip "
                                                    "holds under this load) is that 4-cycle reading measured",
                                     "measured_ceilings_Tabsdiff_s": {
                                         "v_qsad_pk_u16_u8 issue rate, VGPR operands": round(qsad_peak, 2),
                                         "v_sad_u8 issue rate": round(sad_peak, 2),
                                         "qsad strip loop (the kernel's inner loop alone: LDS rows + QSADs, block in SGPRs)": round(loops[0], 2),
                                         "v_sad_u8 loop on four pre-shifted window copies (27 KB LDS per wave)": round(loops[1], 2)},
                                     "frac_of_qsad_strip_loop": round(tabs / loops[0], 5),
                                     "level0_launch": {"ms": round(prof["search_level0_ms"], 4),
                                                       "achieved": round(blocks[0] * (2 * R + 1) ** 2 * block * block / (prof["search_level0_ms"] * 1e-3) / 1e12, 3)}},
                         "sq_counters_level0_launch": pmc_sq_summary(),
                         "xcd_round_robin": xcd_check},
            "search_only": {"value": round(sum(blocks) / (prof["search_ms"] * 1e-3) / 1e6, 3), "unit": "Mblocks/s",
                            "blocks": sum(blocks), "ms": round(prof["search_ms"], 4),
                            "note": "the %d search launches alone (all-level block count), from the eager HIP-event pass" % levels},
            "device_ms": {k: round(val, 4) for k, val in prof.items()},
            # the WHOLE step against the roofline that binds its arithmetic (what main_class.cpp:47-55 times): the searches'
            # abs-diffs of all levels over ms_per_step.  The regulariser adds < 2 % arithmetic and most of the time (latency).
            "step_binding": {"unit": "T abs-diff/s", "absdiffs_per_step": step_absdiff, "ms_per_step": round(ms_step, 4),
                             "achieved": round(step_tabs, 3), "peak": VALU_SAD_PEAK_T, "frac": round(step_tabs / VALU_SAD_PEAK_T, 5),
                             "note": "search abs-diffs of every level / the timed step (search + regulariser + expand); "
                                     "roofline.binding prices the search launches alone"},
        }
        # the regulariser takes most of the step although it is 2 % of the arithmetic: every sweep is a
        # chain of dependent block updates.  Algorithmic bytes of a sweep at block size b: per b x b block
        # its own pixels, nine candidate blocks (10 b^2) and nine MVs in, one out (40).
        reg_bytes = 0
        for l in range(levels):
            b = block
            while b > 1:
                reg_bytes += 2 * ((pw >> l) // b) * ((ph >> l) // b) * (10 * b * b + 40)
                b >>= 1
        reg_gbs = reg_bytes / (prof["regularize_ms"] * 1e-3) / 1e9
        out["regularizer"] = {"bound": "latency (dependent block updates; neither HBM nor VALU)",
                              "kernels": "k_reg_pass1<b> + k_reg_solve<b>, %d sweeps per pyramid" % (2 * levels * (block.bit_length() - 1)),
                              "ms": round(prof["regularize_ms"], 4),
                              "share_of_step": round(prof["regularize_ms"] / prof["total_ms"], 3),
                              "algorithmic_bytes": reg_bytes, "achieved": round(reg_gbs, 2), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(reg_gbs / HBM_PEAK_GBS, 5)}
        if sequence is not None:
            out["sequence"] = sequence
        if sequence_deep is not None:
            out["sequence_deep"] = sequence_deep
        if host_boundary is not None:
            out["host_boundary"] = host_boundary
        out["epe_vs_middlebury_gt"] = epe_on_ground_truth(bbme, local_rank)
        if not use_dist:
            # the opt-in Jacobi regulariser (bbme_set_regularizer_mode; NOT the reference's field): beside `value`, never as it
            mf.set_regularizer_mode(True)
            for _ in range(3):
                mf.estimate_async()
            mf.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                mf.estimate_async()
            mf.synchronize()
            dtj = (time.perf_counter() - t0) / args.steps
            jac_flow = mf.get_flow()
            mf.set_regularizer_mode(False)
            epe_j = epe_on_ground_truth(bbme, local_rank, jacobi=True)
            out["fast_regularizer"] = {"mode": "jacobi (opt-in, not bit-exact: every sweep reads the previous sweep's field)",
                                       "value": round(blocks[0] / dtj / 1e6, 4), "unit": "Mblocks/s",
                                       "ms_per_step": round(dtj * 1e3, 4),
                                       "cells_differing_from_exact": round(float((jac_flow != result_flow).any(-1).mean()), 5),
                                       "epe_vs_middlebury_gt": epe_j["value"] if epe_j else None,
                                       "epe_exact": out["epe_vs_middlebury_gt"]["value"] if out["epe_vs_middlebury_gt"] else None}
        if not use_dist and not args.no_other_workloads:
            # the other single-GPU BASELINE configs (never `value`): the driver's one command times them all
            out["other_workloads"] = {name: other_workload(bbme, torch, name, local_rank, max(4, args.steps // 4), 3,
                                                           not args.no_cpu_baseline)
                                      for name in ("cfg2", "cfg4", "cfg1", "ref", "ref2") if name != args.workload}
        if not args.no_cpu_baseline:
            (dt1, _, par1), (dtn, threads, parn) = cpu_baseline(f1, f2, search, block, levels, result_flow)
            model, nproc, usable, quota_text = cpu_info()
            out["cpu_baseline"] = {"value": round(blocks[0] / dt1 / 1e6, 5), "unit": "Mblocks/s", "cores": 1,
                                   "kind": "port", "seconds": round(dt1, 2),
                                   "sample": "the same full %s pair, whole pyramid, oracle/bbme_oracle.c "
                                             "(-O3 -march=native, 1 thread like the reference, no SAD cache)" % args.workload,
                                   "all_cores": {"value": round(blocks[0] / dtn / 1e6, 5), "unit": "Mblocks/s",
                                                 "cores": threads, "seconds": round(dtn, 2),
                                                 "note": "same pair; OpenMP over the macroblocks of the search "
                                                         "(calcLevelBM), regulariser sweeps sequential"},
                                   "host": {"cpu_model": model, "nproc": nproc, "usable_cores": usable, "cgroup_quota": quota_text}}
            out["parity_vs_oracle"] = bool(par1 and parn)
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out))
        sys.stdout.flush()
    mf.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
