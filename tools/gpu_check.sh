#!/bin/bash
# Quick GPU check of the statistics forms: their tests, both entry points timed alone, the reference-frame path, the frames table.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/check; mkdir -p $O
{
  echo "=== $(date) pytest stats"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "statistics or stats or colormatch or color_match or native_library or adjacent" 2>&1 | tail -5
  echo "=== $(date) forms alone"; timeout 300 python tools/bench_stats.py --libs r04=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so --frames 1,2,4 --rounds 7 --json $O/bench_stats_forms.json 2>&1 | grep "stats\]"
  echo "=== $(date) reference frame"; timeout 300 python tools/bench_refstats.py --libs r04=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so 2>&1 | grep "ref\]"
  echo "=== $(date) frames table"; timeout 600 python tools/frames_table.py 2>&1 | grep "frames\]" | cut -c1-330; cp gpurun_out/frames_table.json $O/frames_table.json 2>/dev/null
  echo "=== $(date) done"
} > $O/check.log 2>&1
cat $O/check.log | cut -c1-400
