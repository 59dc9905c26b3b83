#!/bin/bash
# Development aid: the library built with -DBBME_PHASE_PROFILE (shader-clock stamps in the solver's chain rounds) into
# scratch/libbbme_prof.so; use with BBME_LIB=scratch/libbbme_prof.so python scripts/sweep_timeline.py
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/scratch
C=$REPO/blockbasedmotionestimation_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -shared -DBBME_PHASE_PROFILE -I $REPO/include -I $C -x hip $C/bbme_host.cpp $C/bbme_device.hip -o $REPO/scratch/libbbme_prof.so
