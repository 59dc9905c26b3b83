"""Frame pairs of a sequence sharded one pair per GPU.

Pairs are independent in the reference (an MF object holds all state of one pair,
motion_framework.h:37-46; nothing is carried from pair to pair), so pair p goes to rank
p % world_size and runs the whole pyramid there with no exchange.  The only collective is the
gather of the finished dense .flo fields on rank 0 at the end (torch.distributed: backend
"nccl" is RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
import os

import numpy as np


def shard_pairs(n_pairs, rank, world_size):
    """Global indices of the pairs rank `rank` computes."""
    return list(range(rank, n_pairs, world_size))


def estimate_pairs_pipelined(pairs, search_size, block_size, device=0, in_flight=4):
    """All pairs of `pairs` (a list of (frame1, frame2), equal sizes) on ONE GPU, `in_flight` of them at a time.

    One context per slot, created once (level state, launch graph) and re-used round-robin; every
    context has its own stream, so while one pair's regulariser walks its dependency chains the
    chip works on the others.  Returns the unpadded (H, W, 2) float32 fields in input order.
    The result of a pair does not depend on what else is in flight (tests/test_gpu_parity.py).
    HIP maps streams onto 4 hardware queues by default; with more pairs than that in flight export
    GPU_MAX_HW_QUEUES (e.g. 16) before the process starts the HIP runtime (4 pairs: 21 -> 30 Mblocks/s on cfg3).
    """
    from .motion_framework import MF
    if not pairs:
        return []
    slots = []
    out = [None] * len(pairs)
    pending = []                                           # (slot, pair index), oldest first

    def collect():
        slot, idx = pending.pop(0)
        mf = slots[slot]
        h, w = pairs[idx][0].shape
        flow = mf.get_flow()                               # waits for this context's stream only
        out[idx] = np.ascontiguousarray(flow[mf.padding_y:mf.padding_y + h, mf.padding_x:mf.padding_x + w])

    try:
        for idx, (f1, f2) in enumerate(pairs):
            if len(slots) < max(1, in_flight):
                slots.append(MF(f1, f2, search_size, block_size, len(block_size), device=device))
                slot = len(slots) - 1
            else:
                slot = pending[0][0]
                collect()
                slots[slot].set_frames(f1, f2)
            slots[slot].estimate_async()
            pending.append((slot, idx))
        while pending:
            collect()
    finally:
        for mf in slots:
            mf.close()
    return out


def _gpu_compute(search_size, block_size, device):
    from .motion_framework import MF

    def run(frame1, frame2):
        mf = MF(frame1, frame2, search_size, block_size, len(block_size), device=device)
        try:
            flow = mf.calcMotionBlockMatching()
            py, px = mf.padding_y, mf.padding_x
            h, w = frame1.shape
            return np.ascontiguousarray(flow[py:py + h, px:px + w])
        finally:
            mf.close()
    return run


def estimate_sequence(pairs, search_size, block_size, n_pairs=None, out_dir=None, compute=None,
                      device=0, group=None):
    """Run the local shard and gather every pair's (H, W, 2) float32 field on rank 0.

    pairs    : dict {global_pair_index: (frame1, frame2)} holding at least this rank's shard
    compute  : callable (frame1, frame2) -> unpadded (H, W, 2) float32 flow; default = the HIP path
    Returns the list of all fields in pair order on rank 0 (and writes NNNN.flo files into out_dir
    when given), None on the other ranks.
    """
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    if n_pairs is None:
        n_pairs = max(pairs) + 1
    if compute is None:
        compute = _gpu_compute(search_size, block_size, device)
    mine = shard_pairs(n_pairs, rank, world)
    local = {p: compute(*pairs[p]) for p in mine}
    results = None
    if not distributed or world == 1:
        results = [local[p] for p in range(n_pairs)]
    else:
        on_gpu = dist.get_backend(group) == "nccl"
        dev = torch.device("cuda", device) if on_gpu else torch.device("cpu")
        rounds = (n_pairs + world - 1) // world
        results = [None] * n_pairs if rank == 0 else None
        shape = next(iter(local.values())).shape if local else None
        shapes = [None] * world
        dist.all_gather_object(shapes, shape, group=group)
        shape = next(s for s in shapes if s is not None)
        for k in range(rounds):                           # one gather per round of `world` pairs
            p = k * world + rank
            t = torch.from_numpy(local[p]).to(dev) if p < n_pairs else torch.zeros(shape, dtype=torch.float32, device=dev)
            bucket = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(world)] if rank == 0 else None
            dist.gather(t, bucket, dst=0, group=group)
            if rank == 0:
                for r in range(world):
                    q = k * world + r
                    if q < n_pairs:
                        results[q] = bucket[r].cpu().numpy()
    if rank == 0 and out_dir is not None:
        from .rw_flow import Flow
        os.makedirs(out_dir, exist_ok=True)
        for p, f in enumerate(results):
            Flow().WriteFlowFile(f, os.path.join(out_dir, "%04d.flo" % p))
    return results if rank == 0 else None
