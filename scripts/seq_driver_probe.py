#!/usr/bin/env python3
"""bbme_seq on 4K pairs (one GPU): per-round device phases with and without the speculative graph.  Development aid."""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blockbasedmotionestimation_amd.synth import synth_pair      # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    files = []
    for p in range(n):
        f1, f2, _ = synth_pair(3840, 2160, 1030 + p, max_motion=24)
        for k, f in enumerate((f1, f2)):
            path = os.path.join(d, "p%02d_%d.pgm" % (p, k))
            with open(path, "wb") as fh:
                fh.write(b"P5\n3840 2160\n255\n" + f.tobytes())
            files.append(path)
    for env, writers in (({}, 1), ({}, 3), ({}, 6), ({"BBME_SPECULATE": "0"}, 3)):
        out = os.path.join(d, "out_%s_%d" % ("spec" if not env else "plain", writers))
        os.makedirs(out)
        e = dict(os.environ, **env)
        r = subprocess.run([os.path.join(ROOT, "blockbasedmotionestimation_amd", "bbme_seq"), "--gpus", "1", "--levels", "4", "--block", "16",
                            "--search", "80", "--writers", str(writers), "--out", out] + files, env=e, capture_output=True, text=True, timeout=600)
        print("== %s, %d writer(s): rc=%d" % (env or "default", writers, r.returncode))
        print(r.stdout[-3000:])
        print(r.stderr[-500:])
