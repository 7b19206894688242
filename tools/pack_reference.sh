#!/bin/bash
# Build container only: tar the reference's own Python package (minus its 15 MB of test data) into the git-ignored
# gpurun_scratch/ so that ONE gpurun call can time the real `OvercookedEnv.step` on the GPU box's host cores
# (tools/time_reference_python.py).  The tarball is never committed and nothing in the product, tests or bench reads it.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_scratch
tar -C /root/reference/src --exclude='overcooked_ai_py/data/testing' --exclude='__pycache__' --exclude='*.pkl' \
    -czf $R/gpurun_scratch/ref_src.tgz overcooked_ai_py
ls -la $R/gpurun_scratch/ref_src.tgz
