#!/bin/bash
# bench + kernel statistics + PMC traffic of the four bench workloads only (the fingerprinted files): tools/collect_r06_profiles.sh <tag>
tag=${1:-r06_v4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh ${tag}
bash tools/collect_profiles.sh ${tag}_lattice --lattice 64
bash tools/collect_profiles.sh ${tag}_permute --permute 42
bash tools/collect_profiles.sh ${tag}_random --random 42
