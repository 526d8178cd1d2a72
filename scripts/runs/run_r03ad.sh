#!/bin/bash
# config 1 (reference plan file, bench multi): kernel durations against launch gaps (rocprofv3 --kernel-trace of the compiled driver)
source scripts/gpu_steps.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for prec in z f; do
  rm -rf gpurun_out/c1_$prec
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c1_$prec -- tfqmrgpu_amd/lib/bench_tfqmrgpu multi tests/golden/plan_unordered.14-287-16.gz $prec 20 5 > gpurun_out/c1_$prec.log 2>&1
  f=$(find gpurun_out/c1_$prec -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [r for r in rows if "spmm" in r["Kernel_Name"]]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in k]
g = [int(k[i + 1]["Start_Timestamp"]) - int(k[i]["End_Timestamp"]) for i in range(len(k) - 1)]
g = [x for x in g if x < 100000]
print(k[0]["Kernel_Name"][:90], "launches", len(k), "duration us: min %.1f median %.1f" % (min(d) / 1e3, sorted(d)[len(d) // 2] / 1e3), "| gap us: median %.1f" % (sorted(g)[len(g) // 2] / 1e3),
      "| grid", k[0].get("Grid_Size"), "wg", k[0].get("Workgroup_Size"), "vgpr", k[0].get("VGPR_Count"), "lds", k[0].get("LDS_Block_Size"))
PY
done
