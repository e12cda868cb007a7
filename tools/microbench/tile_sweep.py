#!/usr/bin/env python3
"""Every MFMA tiling of the dense conv kernel on the layer shapes of the R50 SlowFast step (SF_CONV_CFG forces one
tiling per process; -1 = the launcher's own choice).  SF_SPLIT_K=0 keeps the whole-output split-K out of the picture.
usage: tools/microbench/tile_sweep.py            (parent: runs one child per tiling)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CFGS = {-1: "auto", 2: "64x64", 3: "128x64", 4: "128x128", 5: "112x64", 6: "112x128", 7: "224x128"}
K3 = ((1, 3, 3), (1, 1, 1), (0, 1, 1))
T3 = ((3, 1, 1), (1, 1, 1), (1, 0, 0))
SHAPES = [("s2.a 256->64", 8, 8, 56, 56, 256, 64), ("s2.b 3x3 64", 8, 8, 56, 56, 64, 64, *K3),
          ("s2.c 64->256 +res", 8, 8, 56, 56, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), True),
          ("s3.a 512->128", 8, 8, 28, 28, 512, 128), ("s3.b 3x3 128", 8, 8, 28, 28, 128, 128, *K3),
          ("s3.c 128->512 +res", 8, 8, 28, 28, 128, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), True),
          ("s3.a t3 576->256", 8, 8, 28, 28, 576, 256, *T3),
          ("s4.a t3 1024->256", 8, 8, 14, 14, 1024, 256, *T3), ("s4.b 3x3 256", 8, 8, 14, 14, 256, 256, *K3),
          ("s4.c 256->1024 +res", 8, 8, 14, 14, 256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), True),
          ("s5.a t3 2048->512", 8, 8, 7, 7, 2048, 512, *T3), ("s5.b 3x3 512", 8, 8, 7, 7, 512, 512, *K3),
          ("fast s3.b 3x3 64 T32", 8, 32, 14, 14, 64, 64, *K3), ("fast s2.c 8->32... skip", 0)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "efficient-slowfast_amd")]
    import microbench_conv as m
    for sh in SHAPES:
        if len(sh) > 2:
            m.run(*sh)
    sys.exit(0)
for cfg, name in CFGS.items():
    for split in (("0",) if cfg >= 0 else ("0", "1")):
        env = dict(os.environ, SF_CONV_CFG=str(cfg), SF_SPLIT_K=split)
        print("== tiling %s, split-K %s" % (name, split), flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
