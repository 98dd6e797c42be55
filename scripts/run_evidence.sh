#!/bin/bash
# Collects the round's evidence at ONE commit: refuses a dirty tree (profiles at a SHA that HEAD had
# already left), passes the commit to the box, copies what comes back into profiles/rNN/.
#   bash scripts/run_evidence.sh [parts, default abcdef]
set -e
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- parakeet_slam_amd bench.py scripts include oracle tests)" ]; then
  echo "the tree has uncommitted changes: commit first, the evidence is stamped with HEAD"; git status --short | head; exit 2
fi
python -m parakeet_slam_amd.build > /dev/null
python -m parakeet_slam_amd.build --stamps > /dev/null
SHA=$(git rev-parse --short HEAD)
/usr/local/graft/bin/gpurun --timeout 1200 -- "PK_GIT_SHA=$SHA EV_PARTS=${1:-abcdte} bash scripts/gpu_r6_evidence.sh"
echo "copy gpurun_out/r06ev/* into profiles/r06/ (git $SHA)"
