"""Shared helpers of the parity tests: drive the product (HIP, through the C-ABI) and the
oracle (CPU restatement) through the reference's level schedule stage by stage."""
import numpy as np


def oracle_schedule(omf, levels, on_stage=None):
    """MF::calcMotionBlockMatching's loop (motion_framework.cpp:115-206) on the oracle,
    reporting every intermediate MV grid: on_stage(name, level, block, mvs int32 (rows, cols, 2))."""
    L = levels
    for lvl in range(L - 1, -1, -1):
        B = omf.block_size(lvl)
        if lvl != L - 1:
            omf.copy_mvs(lvl)
        omf.calc_level_bm(lvl)
        if on_stage:
            on_stage("search", lvl, B, omf.block_mvs(lvl, B))
        b, lam = B, float(B // 2)
        while b > 1:
            for mult in (1, 2):
                omf.set_block_size(lvl, b)
                omf.set_lambda(lvl, lam)
                omf.regularize_mvs(lvl, mult)
                if on_stage:
                    on_stage("sweep%d" % mult, lvl, b, omf.block_mvs(lvl, b))
            omf.divide_blocks(lvl)
            b >>= 1
            lam *= 2
        omf.set_block_size(lvl, B)
    omf.set_block_size(0, 2)
    omf.copy_to_all_pixels(0)
    return omf.flow(0).copy()


def gpu_schedule(mf, levels, blocks, on_stage=None):
    """The same schedule on the product, one C-ABI stage call at a time."""
    for lvl in range(levels - 1, -1, -1):
        B = blocks[lvl]
        mf.stage_search(lvl)
        if on_stage:
            on_stage("search", lvl, B, mf.stage_get_mvs(lvl, B).astype(np.int32))
        b = B
        while b > 1:
            for mult in (1, 2):
                mf.stage_regularize(lvl, b, mult)
                if on_stage:
                    on_stage("sweep%d" % mult, lvl, b, mf.stage_get_mvs(lvl, b).astype(np.int32))
            b >>= 1
    mf.stage_expand()
    return mf.get_flow()


def compare_stagewise(bbme, oracle, f1, f2, search, block, use_planes=True, raster=False):
    """Runs both sides stage by stage on the same planes; asserts every grid is identical.
    Returns (flow_gpu, flow_oracle)."""
    L = len(block)
    omf = oracle.OracleMF(f1, f2, search, block)
    mf = bbme.MF(f1, f2, search, block, L)
    if raster:
        omf.set_raster_search(True)
        mf.set_search_mode(True)
    assert (mf.padded_width, mf.padded_height, mf.padding_x, mf.padding_y) == \
           (omf.padded_width, omf.padded_height, omf.padding_x, omf.padding_y)
    if use_planes:
        # hand the oracle's planes to the kernels so that pyramid construction (host prep,
        # parity unpinned) cannot leak into hot-path parity
        for lvl in range(L):
            mf.set_level_planes(lvl, omf.image(lvl, 1), omf.image(lvl, 2))
    exp = []
    oflow = oracle_schedule(omf, L, lambda *a: exp.append(a))
    got = []
    gflow = gpu_schedule(mf, L, block, lambda *a: got.append(a))
    assert len(exp) == len(got)
    for (en, el, eb, ev), (gn, gl, gb, gv) in zip(exp, got):
        assert (en, el, eb) == (gn, gl, gb)
        bad = np.argwhere((ev != gv).any(-1))
        assert bad.size == 0, "stage %s level %d block %d: %d of %d MVs differ, first at %s: oracle %s gpu %s" % (
            en, el, eb, len(bad), ev.shape[0] * ev.shape[1], bad[0], ev[tuple(bad[0])], gv[tuple(bad[0])])
    assert np.array_equal(oflow, gflow)
    mf.close()
    omf.close()
    return gflow, oflow
