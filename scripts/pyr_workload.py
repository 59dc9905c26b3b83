#!/usr/bin/env python3
"""MF::MF on the GPU (zero border + pyrDown cascade, bbme_set_frames_device) for 4K frames, a few times: the workload for
`rocprofv3 --kernel-trace --stats` when the padding / pyrDown kernels are timed (host prep on the GPU)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
import blockbasedmotionestimation_amd as bbme                    # noqa: E402

w, h = 3840, 2160
f1, f2, _ = bbme.synth_pair(w, h, 1030, max_motion=24)
t1, t2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
mf = bbme.MF(t1, t2, [80] * 4, [16] * 4, 4, frames_on_device=True)
for _ in range(20):
    mf.set_frames_device(t1, t2)
mf.synchronize()
mf.close()
print("pyr workload done")
