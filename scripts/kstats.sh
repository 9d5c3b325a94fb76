#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace summary of a short bench run -> stdout (top N kernels)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/kt.log 2>&1
python scripts/rocpd_stats.py /tmp/prof_kt/kt_results.db /tmp/kstats.md > /dev/null
head -${1:-30} /tmp/kstats.md
