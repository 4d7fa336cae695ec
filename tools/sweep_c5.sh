#!/bin/bash
# config 5 (1.31 M triangles) through tools/sweep_variants.py: generates the scene like bench.py does, then sweeps the given variants
# usage: tools/sweep_c5.sh "9,30,31" [extra sweep args]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SCENE=$(python3 -c "
import sys; sys.path.insert(0, '$REPO')
import bench, acgpathtracing_amd as pt
print(bench.scene_path(pt, 'stress_1m.obj'))")
python3 $REPO/tools/sweep_variants.py --scene $SCENE --fuse 2 --chunks 0 --rounds ${ROUNDS:-3} --variants "$1" "${@:2}"
