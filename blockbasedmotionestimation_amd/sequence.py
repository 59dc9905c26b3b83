"""Frame pairs of a sequence over GPUs and within one GPU.

Pairs are independent in the reference (an MF object holds all state of one pair,
motion_framework.h:37-46; nothing is carried from pair to pair), so they shard with no exchange:

* across GPUs -- pair p goes to rank p % world_size (one process per GPU) and runs the whole pyramid there; the only
  collective is ONE gather per step of the results on rank 0.  `CellGather` / `mf_cell_gather` is that step as
  `bench.py --gpus N` runs it: the results travel as compact int16 cell grids (1/16 of the dense field), staging buffers are
  double-buffered, and rank 0 expands the gathered grids to the dense .flo fields on a second stream beside the next
  estimate.  (`csrc/seq_schedule.hpp` + `bbme_seq` is the same pipeline in C++ over RCCL, without torch.)
* within a GPU -- `estimate_pairs_pipelined`: the pairs that share a GPU go into BATCHED contexts (`MFBatch`,
  bbme_create_batch: every kernel works on all pairs of a context at once), a few contexts side by side on their own streams.
* `estimate_sequence` is the convenience form on top of any `compute` callable and any torch.distributed backend ("nccl" is
  RCCL over xGMI on ROCm; "gloo" in the CPU tests): it gathers the finished dense fields round by round and optionally writes
  the .flo files.  It is not the timed path.
"""
import os

import numpy as np


def local_device():
    """The GPU of this process: one process per GPU, LOCAL_RANK as torch.distributed.run exports it."""
    return int(os.environ.get("LOCAL_RANK", "0"))


def cells_to_words(cells):
    """(rows, cols, 2) int16 (dx, dy) per 2x2 cell -> (rows, cols) int32 words, dx in the low half: the layout of the
    MV grids in HBM and the unit the gather moves (NCCL has no int16)."""
    c = np.ascontiguousarray(cells, np.int16)
    return c.view(np.int32).reshape(c.shape[0], c.shape[1])


def flow_to_cells(flow_padded):
    """The dense padded field -> its 2x2-cell grid (the field is constant on 2x2 cells, motion_framework.cpp:205-206)."""
    return np.ascontiguousarray(flow_padded[::2, ::2]).astype(np.int16)


def expand_cells_host(words):
    """copy_to_all_pixels (motion_framework.cpp:815-826) on the host: (rows, cols) int32 words -> dense
    (2*rows, 2*cols, 2) float32.  The CPU stand-in for bbme_expand_cells_device in the gloo tests."""
    w = np.ascontiguousarray(words, np.int32)
    mv = w.view(np.int16).reshape(w.shape[0], w.shape[1], 2).astype(np.float32)
    return np.repeat(np.repeat(mv, 2, axis=0), 2, axis=1)


class CellGather:
    """The multi-GPU step of a sequence (BASELINE configs[4]): every rank estimates one pair per step, the
    results travel as compact cell grids -- one packed int16 (dx, dy) pair per 2x2 cell, 16x smaller than the dense
    field and exactly the same information -- in ONE gather to rank 0 (torch.distributed: "nccl" is RCCL over xGMI;
    "gloo" in the CPU tests), and rank 0 expands every gathered grid to the dense .flo field.

    estimate()                 enqueues this rank's estimate (GPU: on the current stream, no host wait; CPU: computes)
    cells                      torch int32 tensor (rows, cols) the estimate leaves its result in
    expand(words, flow, strm)  rank 0: one gathered grid -> dense (2*rows, 2*cols, 2) float32 tensor `flow`;
                               strm = raw HIP stream handle the expansion must run on (None on the CPU)
    On a GPU, with `overlap` (default), the gather of step i and rank 0's expansions run on a second stream beside the
    estimate of step i + 1: each step copies its grid into one of two staging buffers; events order the two streams.
    With overlap=False the estimate, the gather and the expansions are simply enqueued in order on the work stream.  That is
    the better form when the estimate is a hipGraph with a forked branch (the speculative search): a second stream waiting
    for an event recorded behind such a graph costs the next replay about 0.7 ms on ROCm 7 (scripts/dist_step_probe.py:
    1.75 -> 2.45 ms per 4K step), far more than the ~0.1 ms of gather and expansions it would hide.  bench.py measures
    both pairings before the timed region and keeps the faster.  `flows` (rank 0) holds the dense fields of the last
    finished step, one per rank.
    """

    def __init__(self, estimate, cells, expand, group=None, dst=0, overlap=True):
        import torch
        import torch.distributed as dist
        self._dist, self._torch = dist, torch
        self.estimate, self.cells, self.expand = estimate, cells, expand
        self.group, self.dst, self.overlap = group, dst, overlap
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.on_gpu = cells.is_cuda
        rows, cols = cells.shape
        root = self.rank == dst
        self.gather_list = [torch.empty_like(cells) for _ in range(self.world)] if root else None
        self.flows = torch.empty((self.world, 2 * rows, 2 * cols, 2), dtype=torch.float32, device=cells.device) if root else None
        self.stage = [torch.empty_like(cells) for _ in range(2)]
        self.steps = 0
        if self.on_gpu:
            self.work_stream = torch.cuda.current_stream(cells.device)
            self.side_stream = torch.cuda.Stream(device=cells.device)
            self.ev_ready = [torch.cuda.Event() for _ in range(2)]
            self.ev_free = [torch.cuda.Event() for _ in range(2)]

    def step(self):
        dist, torch = self._dist, self._torch
        b = self.steps & 1
        self.steps += 1
        if not self.on_gpu:
            self.estimate()
            self.stage[b].copy_(self.cells)
            dist.gather(self.stage[b], self.gather_list, dst=self.dst, group=self.group)
            if self.rank == self.dst:
                for r in range(self.world):
                    self.expand(self.gather_list[r], self.flows[r], None)
            return
        if not self.overlap:
            self.estimate()
            with torch.cuda.stream(self.work_stream):
                dist.gather(self.cells, self.gather_list, dst=self.dst, group=self.group)
                if self.rank == self.dst:
                    for r in range(self.world):
                        self.expand(self.gather_list[r], self.flows[r], self.work_stream.cuda_stream)
            return
        self.work_stream.wait_event(self.ev_free[b])      # the gather that read this staging buffer two steps ago is done
        self.estimate()
        with torch.cuda.stream(self.work_stream):
            self.stage[b].copy_(self.cells)
        self.ev_ready[b].record(self.work_stream)
        with torch.cuda.stream(self.side_stream):
            self.side_stream.wait_event(self.ev_ready[b])
            dist.gather(self.stage[b], self.gather_list, dst=self.dst, group=self.group)
            if self.rank == self.dst:
                for r in range(self.world):
                    self.expand(self.gather_list[r], self.flows[r], self.side_stream.cuda_stream)
            self.ev_free[b].record(self.side_stream)

    def fence(self):
        """Both streams idle on every rank."""
        if self.on_gpu:
            self.side_stream.synchronize()
            self.work_stream.synchronize()
        self._dist.barrier(group=self.group)


def mf_cell_gather(mf, device, group=None, overlap=True):
    """CellGather over a context (MF) whose frames are set: the estimate, the context's cell grid and
    bbme_expand_cells_device_on.  The context is moved onto torch's current stream, which must not be the default
    stream (handle 0 means "private stream" to bbme_set_stream)."""
    import torch
    stream = torch.cuda.current_stream(device)
    if stream.cuda_stream == 0:
        raise ValueError("mf_cell_gather: run under an explicit torch.cuda.Stream (the default stream has handle 0)")
    mf.set_stream(stream.cuda_stream)

    class _View:
        pass
    v = _View()
    v.__cuda_array_interface__ = {"shape": (mf.padded_height // 2, mf.padded_width // 2), "typestr": "<i4",
                                  "data": (mf.cells_device_ptr(), False), "version": 2, "strides": None}
    cells = torch.as_tensor(v, device=torch.device("cuda", device))

    def expand(words, flow, strm):
        mf.expand_cells_device(words.data_ptr(), flow.data_ptr(), strm)
    return CellGather(mf.estimate_async, cells, expand, group=group, overlap=overlap)


def shard_pairs(n_pairs, rank, world_size):
    """Global indices of the pairs rank `rank` computes."""
    return list(range(rank, n_pairs, world_size))


def estimate_pairs_pipelined(pairs, search_size, block_size, device=None, in_flight=4, batch=2):
    """All pairs of `pairs` (a list of (frame1, frame2), equal sizes) on ONE GPU, `in_flight` of them at a time.

    The pairs go, `batch` at a time, into batched contexts (MFBatch: one launch sequence for all pairs of the context, the
    pair is a grid dimension), in_flight // batch contexts created once (level state, launch graph) and re-used round-robin,
    each on its own stream: while one context's regulariser walks its dependency chains the chip searches for another.
    Why batches: the device dispatches the dependent kernels of many streams no faster than one per ~4 us chip-wide, so one
    context per pair is dispatch-bound at 69 launches per pair (8 pairs at 4K: 34 Mblocks/s with 8 contexts x 1 pair,
    45 with 4 x 2).  Returns the unpadded (H, W, 2) float32 fields in input order; a pair's result does not depend on
    what shares its context or the GPU (tests/test_gpu_parity.py).  More than 4 streams: export GPU_MAX_HW_QUEUES (e.g. 16)
    before the process starts the HIP runtime.
    """
    from .motion_framework import MFBatch
    if not pairs:
        return []
    if device is None:
        device = local_device()
    per = max(1, min(batch, in_flight, len(pairs)))
    n_slots = max(1, in_flight // per)
    groups = [list(range(i, min(i + per, len(pairs)))) for i in range(0, len(pairs), per)]
    slots = []
    out = [None] * len(pairs)
    pending = []                                           # (slot, pair indices), oldest first

    def collect():
        slot, idxs = pending.pop(0)
        mf = slots[slot]
        for p, idx in enumerate(idxs):
            h, w = pairs[idx][0].shape
            flow = mf.get_pair_flow(p)                     # waits for this context's stream only
            out[idx] = np.ascontiguousarray(flow[mf.padding_y:mf.padding_y + h, mf.padding_x:mf.padding_x + w])

    try:
        for idxs in groups:
            frames = [pairs[i] for i in idxs] + [pairs[idxs[-1]]] * (per - len(idxs))     # a short last group: padded, not read
            if len(slots) < n_slots:
                slots.append(MFBatch(frames, search_size, block_size, len(block_size), device=device))
                slot = len(slots) - 1
                if n_slots * per > 1:
                    slots[slot].set_speculation(False)      # the other pairs in flight fill the chip already
            else:
                slot = pending[0][0]
                collect()
                for p, (f1, f2) in enumerate(frames):
                    slots[slot].set_pair(p, f1, f2)
            slots[slot].estimate_async()
            pending.append((slot, idxs))
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
                      device=None, group=None):
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
    if device is None:
        device = local_device()                           # one process per GPU: never every rank on cuda:0
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
