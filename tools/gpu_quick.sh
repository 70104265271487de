#!/bin/bash
# Quick GPU iteration: parity tests + diagnostics micro-benchmarks.  bash tools/gpu_quick.sh <tag> [pytest -k expr] [diag --match]
TAG=${1:-q}
mkdir -p gpurun_out
{
  echo "=== $(date) pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider ${2:+-k "$2"} 2>&1 | tail -60
  echo "=== $(date) diag"; timeout 900 python tools/gpu_diag.py --frames 16 --iters 5 ${3:+--match "$3"} --out gpurun_out/diag_${TAG}.json 2>&1 | grep -v amdgpu.ids
  echo "=== $(date) done"
} > gpurun_out/quick_${TAG}.log 2>&1
tail -3 gpurun_out/quick_${TAG}.log
