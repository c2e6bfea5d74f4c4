#!/bin/bash
# tools/class_trace.sh [N] -- the class-per-pair driver (the reference's main.cpp shape) on N short pairs: elapsed time and the engine's phase timings
N=${1:-4000}
python - <<PY
import sys; sys.path.insert(0, ".")
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch
dpx.write_pairs_file(make_ragged_batch($N, 80, 130, 100, 160, seed=6), "/tmp/class_pairs.txt")
PY
make -s -C dpx_gpu_genomics_project_amd/hostcpp
M=dpx_gpu_genomics_project_amd/hostcpp/dpx_class_main
for i in 1 2 3; do $M -pairs /tmp/class_pairs.txt -match 3 -mismatch -1 -open -2 -algo LSW > /tmp/class_out.txt; grep -E "^Elapsed" /tmp/class_out.txt; done
DPX_TRACE=1 $M -pairs /tmp/class_pairs.txt -match 3 -mismatch -1 -open -2 -algo LSW 2> /tmp/class_trace.txt > /dev/null
python3 - <<PY
import re, collections
acc = collections.OrderedDict(); cnt = collections.Counter()
for l in open("/tmp/class_trace.txt"):
    m = re.match(r"\[dpx\] (.*?)\s+([0-9.]+) ms", l)
    if m: acc[m.group(1)] = acc.get(m.group(1), 0.0) + float(m.group(2)); cnt[m.group(1)] += 1
for k, v in acc.items(): print(f"{k:44s} {v:9.2f} ms total  {cnt[k]:5d} x  {1e3 * v / cnt[k]:8.1f} us")
PY
