#!/bin/bash
# effective shader clock per kernel of one tools/kbench.py invocation: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
# (MI355X_MICROARCH.md, DVFS give-back; reads high on dispatches shorter than ~0.3 ms):  bash tools/kclk.sh <tag> <kbench args...>
set -e -o pipefail
tag=${1:?tag}; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/kclk_$tag" -o k -- python3 "$root/tools/kbench.py" "$@" > "$out/kclk_$tag.log" 2>&1
cd "$root"
python - "$out/kclk_$tag" <<'PY' > "$out/kclk_$tag.txt"
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
dur = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur and "vivim" in r["Kernel_Name"]:
            name, ns = dur[r["Dispatch_Id"]]
            acc[name].append((float(r["Counter_Value"]) / 8.0 / ns, ns))
for k in sorted(acc):
    v = acc[k]
    print("%-100s n %3d  avg %8.1f us  clock %.2f GHz" % (k[:100], len(v), sum(x[1] for x in v) / len(v) / 1e3, sum(x[0] for x in v) / len(v)))
PY
rm -rf "$out/kclk_$tag"
cat "$out/kclk_$tag.txt"
