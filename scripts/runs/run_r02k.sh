#!/bin/bash
source scripts/gpu_steps.sh
for wl in stencil3d_32x32_c stencil2d_8x8_z; do
  step 400 cfg_$wl.json python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step 300 pmc_c3.log rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/r02_pmc_c3 -- python3 scripts/pmc_driver.py stencil3d_32x32_c 3
step 300 pmc_c3b.log rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d gpurun_out/r02_pmc_c3b -- python3 scripts/pmc_driver.py stencil3d_32x32_c 3
python3 - <<'PY'
import json,csv,glob,collections
for wl in ("stencil3d_32x32_c","stencil2d_8x8_z"):
    d=json.loads([l for l in open("gpurun_out/cfg_%s.json"%wl) if l.startswith("{")][-1])
    print(wl, d["value"], d["ms_per_step"], d["iterations_per_solve"])
    for k in ("roofline","roofline_multiply","roofline_multiply_native_api","roofline_iteration"):
        r=d[k]; print("  ",k, r.get("kernel","")[:40], r.get("avg_ms", r.get("ms_per_iteration")), r["achieved"], r["unit"], r["frac"])
for dd in ("r02_pmc_c3","r02_pmc_c3b"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv"%dd):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            if "spmm" in k: print(dd,k,{c:round(max(x)) for c,x in v.items()})
PY
