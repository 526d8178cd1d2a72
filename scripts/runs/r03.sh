#!/bin/bash
# Every GPU call of round 3 as one parametrised script: gpurun --timeout N -- bash scripts/runs/r03.sh <step>.  A step is what was one
# gpurun call; its comment says what it measured, profiles/r03_*.txt hold what came out (the records name the steps as "r03.sh <step>").
source scripts/gpu_steps.sh
case "$1" in
a)
  # round 3, first call: baseline bench on this round's box + the row-order probe
  step 500 r03a_bench.json python bench.py --steps 5 --warmup 2
  step 900 r03a_row_order.txt python scripts/row_order_probe.py 4,8,12,16,24
  ;;
b)
  # round 3: PMC of the fused multiplies under different row orders / column-group sizes; the new P2 test; bench with the two byte models
  step 400 r03b_p2test.log python -m pytest tests/test_gpu_configs.py -q -k "config2" -x
  step 400 r03b_bench.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  rm -rf gpurun_out/r03b_pmc
  step 600 r03b_pmc.log rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/r03b_pmc -- python3 scripts/row_order_probe.py 4,8,16,49 natural,raster,strip8,tile4 0
  python3 scripts/pmc_by_kernel.py gpurun_out/r03b_pmc k_spmm_ilv16 14 > gpurun_out/r03b_pmc_summary.txt 2>&1
  rm -rf gpurun_out/r03b_pmc
  cat gpurun_out/r03b_pmc_summary.txt | cut -c1-200
  ;;
c)
  # round 3: exact vmcnt waits in k_spmm_ilv16 (no conditional load in the product loop): A/B against the r02 kernel
  step 600 r03c_ab.txt python scripts/ab_fused.py fd2d_16x16_z scripts/bin/libtfQMRgpu_r02.so default scripts/bin/libtfQMRgpu_tails.so scripts/bin/libtfQMRgpu_r02.so default
  cat gpurun_out/r03c_ab.txt
  step 900 r03c_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py -q -x
  ;;
d)
  # round 3: mixed precision 'm', frozen switches (lab build), 3M opt-in: the whole GPU suite, mixed tests first
  step 900 r03d_mixed.log python -m pytest tests/test_gpu_mixed.py -q -x
  tail -30 gpurun_out/r03d_mixed.log
  step 1100 r03d_pytest.log python -m pytest tests -m gpu -q --deselect tests/test_gpu_mixed.py
  grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r03d_pytest.log | tail -15
  ;;
e)
  step 900 r03e_mixed.log python -m pytest tests/test_gpu_mixed.py tests/test_bench_launcher.py tests/test_bench_binary.py -q
  tail -40 gpurun_out/r03e_mixed.log
  ;;
f)
  # round 3: bench with the mixed-precision solve beside the headline; small-system latency baseline; config 3
  step 600 r03f_bench.json python bench.py --steps 5 --warmup 2
  python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03f_bench.json") if l.startswith("{")][-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("kernel", "avg_ms", "frac", "frac_model", "traffic_over_moved")})
print("mixed", d["mixed_precision"])
print("multiply", d["roofline_multiply"]["avg_ms"], d["roofline_multiply"]["frac"], "hbm corner", d["roofline_multiply_hbm_bound"]["frac"])
PY
  step 300 r03f_small.txt python scripts/small_latency.py
  cat gpurun_out/r03f_small.txt
  step 300 r03f_cfg3.json python bench.py --workload stencil3d_32x32_c --steps 20 --warmup 10 --no-cpu-baseline
  tail -c 1500 gpurun_out/r03f_cfg3.json
  ;;
g)
  # round 3: column operations folded into the producers' tails for small systems
  step 300 r03g_small.txt python scripts/small_latency.py
  cat gpurun_out/r03g_small.txt | grep -v amdgpu
  export TFQMRGPU_LIB=$PWD/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  TFQMRGPU_FOLD_MAX=0 timeout 300 python scripts/small_latency.py 2>&1 | grep -v amdgpu | sed 's/^/nofold: /'
  for fm in 0 100000; do
    echo "config 3, FOLD_MAX=$fm"; TFQMRGPU_FOLD_MAX=$fm timeout 300 python bench.py --workload stencil3d_32x32_c --steps 20 --warmup 10 --no-cpu-baseline --no-mixed 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline_iteration']['ms_per_iteration'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
  done
  unset TFQMRGPU_LIB
  step 1100 r03g_pytest.log python -m pytest tests -m gpu -q -x
  grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r03g_pytest.log | tail -15
  ;;
h)
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  rm -rf gpurun_out/r03h_trace
  step 300 r03h_trace.log rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r03h_trace -- python3 scripts/small_one.py
  python3 scripts/trace_gaps.py gpurun_out/r03h_trace 60 > gpurun_out/r03h_gaps.txt 2>&1
  rm -rf gpurun_out/r03h_trace
  cat gpurun_out/r03h_gaps.txt
  step 600 r03h_dev.log python -m pytest tests/test_gpu_parity.py -q -k "device_arrays"
  tail -5 gpurun_out/r03h_dev.log
  ;;
i)
  # chunk length and the fused multiplies (lab build): does a wave with several Y blocks per work group run the multiplies faster?
  for kib in 16 32 64 128; do
    echo "CHUNK_KIB=$kib"
    TFQMRGPU_CHUNK_KIB=$kib timeout 300 python scripts/ab_fused.py fd2d_16x16_z tfqmrgpu_amd/lib/libtfQMRgpu_lab.so 2>&1 | grep -v amdgpu
  done
  ;;
j)
  # column-group size of the launch order where A is large against one X column (config 5: 8x8 z, 8 columns; config 3: 2 columns)
  for g in 4 8 2; do
    echo "stencil2d_8x8_z ORDER_G=$g"
    TFQMRGPU_ORDER_G=$g timeout 300 python scripts/ab_fused.py stencil2d_8x8_z tfqmrgpu_amd/lib/libtfQMRgpu_lab.so 2>&1 | grep -v amdgpu
  done
  for g in 4 8 16; do
    echo "st:16:16:z:128:128:32 (config 4 shard) ORDER_G=$g"
    TFQMRGPU_ORDER_G=$g timeout 300 python scripts/ab_fused.py st:16:16:z:128:128:32 tfqmrgpu_amd/lib/libtfQMRgpu_lab.so 2>&1 | grep -v amdgpu
  done
  ;;
k)
  # work-group descriptors for k_spmm_ilv16 (one scalar load instead of a chain of dependent ones): A/B in the lab build, parity
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for d in 0 1 0 1; do
    echo "DESC=$d"; TFQMRGPU_DESC=$d timeout 300 python scripts/ab_fused.py fd2d_16x16_z $L 2>&1 | grep -v amdgpu
  done
  echo "small systems, DESC=0 then 1"
  TFQMRGPU_LIB=$PWD/$L TFQMRGPU_DESC=0 timeout 300 python scripts/small_latency.py 2>&1 | grep -v amdgpu
  TFQMRGPU_LIB=$PWD/$L TFQMRGPU_DESC=1 timeout 300 python scripts/small_latency.py 2>&1 | grep -v amdgpu
  step 900 r03k_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py tests/test_gpu_configs.py -q -x
  tail -4 gpurun_out/r03k_pytest.log
  ;;
l)
  step 300 r03l_clock.txt python scripts/clock_under_load.py
  grep -v amdgpu gpurun_out/r03l_clock.txt
  ;;
m)
  # round 3: fresh PMC passes of the bench workload (profiles/r03_pmc_summary.json, pmc_traffic.json), mixed-precision ranks test
  step 600 r03m_ranks.log python -m pytest tests/test_gpu_ranks.py tests/test_gpu_mixed.py -q
  tail -3 gpurun_out/r03m_ranks.log
  rm -rf gpurun_out/pmc
  bash scripts/pmc_collect.sh gpurun_out/pmc fd2d_16x16_z
  python3 scripts/pmc_summary.py gpurun_out/pmc > gpurun_out/r03m_pmc_summary.json
  find gpurun_out/pmc -name "*.csv" -delete
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03m_pmc_summary.json"))
for k, v in d.items():
    if "spmm" in k or "x_v6" in k or "xpay" in k or "v5_nrm" in k:
        print(k[:60], v.get("avg_us_working"), v.get("hbm_read_MB(2x FETCH_SIZE)"), v.get("hbm_write_MB"), v.get("l2_hit_rate"), v.get("mfma_busy_per_sq_busy"))
PY
  ;;
n)
  # 8 x 32 and 8 x 64 complex<double> on the row-pair-interleaved order (k_spmm_ilv8w): parity, then A/B against the native order (lab: TFQMRGPU_ILV=16)
  step 900 r03n_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py tests/test_gpu_hash_mode.py -q -x
  tail -4 gpurun_out/r03n_pytest.log
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in st:8:32:z:181:181:4 st:8:64:z:128:128:4; do
    for ilv in 16 1 16 1; do
      echo "$wl TFQMRGPU_ILV=$ilv"; TFQMRGPU_ILV=$ilv timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu
    done
  done
  ;;
o)
  # do X-shaped vectors that lie exactly 2^k bytes apart hurt?  gap behind each vector (lab: TFQMRGPU_SKEW) on the power-of-two configurations
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in st:8:64:z:128:128:4 stencil2d_8x8_z st:16:16:z:128:128:32 fd2d_16x16_z; do
    for sk in 0 4352 69888 1118464; do
      echo "$wl TFQMRGPU_SKEW=$sk"; TFQMRGPU_SKEW=$sk timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu
    done
  done
  ;;
p)
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in st:8:64:z:128:128:4 st:8:64:z:127:127:4; do
    for ilv in 16 1 16 1; do
      echo "$wl TFQMRGPU_ILV=$ilv"; TFQMRGPU_ILV=$ilv timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu
    done
  done
  echo "fd2d_16x16_z: lab against product"
  timeout 300 python scripts/ab_fused.py fd2d_16x16_z $L default 2>&1 | grep -v amdgpu
  ;;
q)
  # where do the small-block shapes stand? (product build; ~0.5 GB per vector in z, 0.25 GB in c)
  for wl in st:8:8:c:362:362:4 st:8:32:c:181:181:4 st:8:64:c:128:128:4 st:8:9:z:241:241:4 st:8:10:z:228:228:4 st:4:4:z:512:512:4 st:4:8:z:362:362:4 st:4:32:z:181:181:4 st:4:4:c:724:724:4 st:4:32:c:256:256:4; do
    echo "$wl"; timeout 300 python scripts/ab_fused.py $wl default 2>&1 | grep -v amdgpu
  done
  ;;
r)
  # 8 x 8 | 32 | 64 complex<float> on the quad-interleaved order (k_spmm_ilv8f): parity, then A/B against the native order (lab: TFQMRGPU_ILV=3)
  step 900 r03r_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py tests/test_gpu_hash_mode.py -q -x
  tail -12 gpurun_out/r03r_pytest.log
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in st:8:8:c:362:362:4 st:8:32:c:181:181:4 st:8:64:c:128:128:4; do
    for ilv in 3 1 3 1; do
      echo "$wl TFQMRGPU_ILV=$ilv"; TFQMRGPU_ILV=$ilv timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu
    done
  done
  ;;
s)
  step 600 r03s_mixed.log python -m pytest tests/test_gpu_mixed.py tests/test_gpu_ranks.py -q
  tail -3 gpurun_out/r03s_mixed.log
  step 300 r03s_bench.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-multiply
  python3 -c "
  import json
  d=json.loads([l for l in open('gpurun_out/r03s_bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['mixed_precision'])"
  ;;
t)
  # config 3 (2 block columns x 1024 chunks): chunk length against the column kernels (lab build)
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for kib in 16 64 128 256; do
    echo "stencil3d_32x32_c CHUNK_KIB=$kib"; TFQMRGPU_CHUNK_KIB=$kib timeout 300 python scripts/ab_fused.py stencil3d_32x32_c $L 2>&1 | grep -v amdgpu
  done
  step 300 r03t_launcher.log python -m pytest tests/test_bench_launcher.py -q
  tail -3 gpurun_out/r03t_launcher.log
  ;;
u)
  # config 5 (8x8 z, 256 x 256 grid, 8 columns): caller-side tiled row numbering (library unchanged): does locality pay where the multiply may be fabric-bound?
  step 300 r03u_tiles4.txt python scripts/tiled_rows_probe.py 4 8
  step 300 r03u_tiles16.txt python scripts/tiled_rows_probe.py 16 8
  grep -v amdgpu gpurun_out/r03u_tiles4.txt; grep -v amdgpu gpurun_out/r03u_tiles16.txt
  ;;
v)
  # soak: repeated solves give identical bits -- small system (column operations folded into the producers' tails), P2 in mixed precision, P2 in z
  step 300 r03v_soak_small.txt python scripts/soak.py fd2d_16x16_z_small 600
  step 300 r03v_soak_small_m.txt python scripts/soak.py fd2d_16x16_z_small 300 m
  step 400 r03v_soak_m.txt python scripts/soak.py fd2d_16x16_z 60 m
  step 400 r03v_soak_z.txt python scripts/soak.py fd2d_16x16_z 200
  for f in small small_m m z; do tail -1 gpurun_out/r03v_soak_$f.txt; done
  step 300 r03v_drv.log python -m pytest tests/test_bench_binary.py -q
  tail -3 gpurun_out/r03v_drv.log
  ;;
w)
  for wl in st:4:4:z:512:512:4 st:4:4:c:724:724:4; do echo "$wl"; timeout 300 python scripts/ab_fused.py $wl default 2>&1 | grep -v amdgpu; done
  step 900 r03w_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py tests/test_gpu_operator.py -q -x
  tail -3 gpurun_out/r03w_pytest.log
  ;;
x)
  # what the four-product default costs the z shapes above 16x16 (lab: TFQMRGPU_3M=0 | 1)
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in st:32:32:z:64:64:8 st:64:64:z:32:32:8 st:16:32:z:90:90:8; do
    for m in 0 1; do echo "$wl TFQMRGPU_3M=$m"; TFQMRGPU_3M=$m timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu; done
  done
  ;;
y)
  # k_spmm_ilv16 forced to four waves per SIMD (launch bound 4 work groups per CU: 128 VGPRs, 24-104 bytes of scratch) against the product
  timeout 600 python scripts/ab_fused.py fd2d_16x16_z tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_w4.so tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_w4.so 2>&1 | grep -v amdgpu
  ;;
z)
  # occupancy probe of the fused 16x16 z multiplies: (a) two waves per SIMD through an unused 60 KiB of dynamic LDS per work group (lab switch),
  # (b) epilogue operands through LDS-DMA (EPI 2: 128 VGPRs = four waves per SIMD, 24 bytes of scratch; EPI 1 stays at 148 = three waves)
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for k in 0 60 0 60; do echo "TFQMRGPU_ILV16_LDS_KIB=$k"; TFQMRGPU_ILV16_LDS_KIB=$k timeout 300 python scripts/ab_fused.py fd2d_16x16_z $L 2>&1 | grep -v amdgpu; done
  timeout 600 python scripts/ab_fused.py fd2d_16x16_z tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_elds.so tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_elds.so 2>&1 | grep -v amdgpu
  ;;
aa)
  # timing-only probes of k_spmm_ilv16 (TFQ_PROBE bits: 1 no products, 2 linear chunk order, 4 plain epilogue accesses, 8 no record reduction,
  # 16 Y not stored, 32 no hash), 30 iterations each; results of these builds are wrong by construction
  export AB_MAXIT=30
  libs="tfqmrgpu_amd/lib/libtfQMRgpu.so"
  for b in 1 3 5 9 17 33 2 4 8 16 32 7; do libs="$libs scripts/bin/libtfQMRgpu_p$b.so"; done
  timeout 800 python scripts/ab_fused.py fd2d_16x16_z $libs tfqmrgpu_amd/lib/libtfQMRgpu.so 2>&1 | grep -v amdgpu
  ;;
ab)
  # timing-only probes: 3 of 4 A fetches skipped (64), 3 of 4 X fetches skipped (128), both (192) -- what sharing operands between block products would buy
  export AB_MAXIT=30
  timeout 800 python scripts/ab_fused.py fd2d_16x16_z tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_p64.so scripts/bin/libtfQMRgpu_p128.so scripts/bin/libtfQMRgpu_p192.so tfqmrgpu_amd/lib/libtfQMRgpu.so 2>&1 | grep -v amdgpu
  ;;
ac)
  # round-3 shape survey: all 15 block shapes x z, c (scripts/shape_survey.sh), to compare with profiles/r01_shape_survey.txt
  timeout 1100 bash scripts/shape_survey.sh > gpurun_out/r03_shape_survey.txt 2>&1
  wc -l gpurun_out/r03_shape_survey.txt
  ;;
ad)
  # config 1 (reference plan file, bench multi): kernel durations against launch gaps (rocprofv3 --kernel-trace of the compiled driver)
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
  ;;
ae)
  # native-API multiply (plain mode of k_spmm_mfma): contiguous eighths of the caller's Y blocks per XCD (lab switch TFQMRGPU_PLAIN_XCD) on config 1 and on P2
  export TFQMRGPU_LIB=$PWD/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for x in 0 1 0 1; do
    echo "TFQMRGPU_PLAIN_XCD=$x"
    for prec in f z; do TFQMRGPU_PLAIN_XCD=$x python -m tfqmrgpu_amd.bench_tfqmrgpu multi tests/golden/plan_unordered.14-287-16.gz $prec 20 5 2>&1 | grep "GPU performance\|maxdev"; done
  done
  for x in 0 1; do echo "P2 TFQMRGPU_PLAIN_XCD=$x"; TFQMRGPU_PLAIN_XCD=$x timeout 300 python scripts/bench_multiply.py fd2d_16x16_z 5 2>&1 | grep -E "^multiply"; done
  ;;
af)
  # native-API multiply: round-robin work groups against contiguous eighths per XCD (TFQMRGPU_PLAIN_XCD) over sizes and shapes
  export TFQMRGPU_LIB=$PWD/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in st:16:16:z:24:24:4 st:16:16:z:48:48:4 st:16:16:z:128:128:4 st:16:16:z:64:64:32 st:16:16:c:48:48:8 st:16:16:c:181:181:4 st:32:32:z:64:64:4 st:32:32:c:32:32:4 st:64:64:z:16:16:4 fd2d_16x16_z_small; do
    for x in 0 1; do echo -n "$wl PLAIN_XCD=$x "; TFQMRGPU_PLAIN_XCD=$x timeout 300 python scripts/bench_multiply.py $wl 5 2>&1 | grep -E "^multiply" | cut -c1-200; done
  done
  ;;
ag)
  # column sums with 32 records in flight per lane (kernels of their own): parity + bit-identity tests, config 3, P2, small systems
  step 900 pytest_r03ag.log python -m pytest tests/test_gpu_hash_mode.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_mixed.py tests/test_gpu_ranks.py -q -x
  tail -3 gpurun_out/pytest_r03ag.log
  python bench.py --workload stencil3d_32x32_c --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03ag_c3.json
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r03ag_c3.json"))
print("config 3: ms_per_step", d["ms_per_step"], "value", d["value"], "iteration", d["roofline_iteration"]["ms_per_iteration"])
print({k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mixed 2>/dev/null | tail -1 > gpurun_out/r03ag_p2.json
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r03ag_p2.json"))
print("P2: ms_per_step", d["ms_per_step"], "value", d["value"])
print({k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
  python scripts/small_latency.py 2>&1 | grep -v amdgpu | tail -6
  ;;
ah)
  # where does the fold of the column operations stop paying (Plan::foldOk = chunks <= 512 was set between two data points)
  timeout 800 python scripts/fold_crossover.py 2>&1 | grep -v amdgpu
  ;;
ai)
  # fold crossover, small end: 16 ... 256 chunks
  timeout 800 python scripts/fold_crossover.py st:16:16:z:8:8:1 st:16:16:z:8:8:2 st:16:16:z:12:12:2 st:16:16:z:16:16:2 st:16:16:z:16:16:4 st:8:8:z:8:8:4 st:8:8:z:16:16:4 st:8:8:z:24:24:4 st:32:32:c:8:8:2 st:4:4:z:32:32:4 FD:1.75,6.75,2,3,0.0,4 2>&1 | grep -v amdgpu
  ;;
aj)
  # fold crossover on a created stream (scripts/small_latency.py's case) and on the null stream
  for st in created null; do
  FOLD_STREAM=$st timeout 500 python scripts/fold_crossover.py st:16:16:z:8:8:1 st:16:16:z:8:8:2 st:16:16:z:16:16:2 st:16:16:z:16:16:4 st:8:8:z:16:16:4 st:8:8:z:24:24:4 FD:1.75,6.75,2,3,0.0,4 FD:6,24,4,2,-0.25,4 st:16:16:z:32:32:4 2>&1 | grep -v amdgpu
  done
  ;;
ak)
  # the stopping decision in the last work group of k_decT<FINAL> / k_probe_col (one rank, unfolded plans): tests, then small-system and P2 timing
  step 900 pytest_r03ak.log python -m pytest tests/test_gpu_hash_mode.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_operator.py tests/test_gpu_ranks.py tests/test_gpu_mixed.py -q -x
  tail -3 gpurun_out/pytest_r03ak.log
  export TFQMRGPU_LIB=$PWD/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for j in 0 1 0 1; do echo "TFQMRGPU_JOIN=$j"; TFQMRGPU_JOIN=$j python scripts/small_latency.py 2>&1 | grep -v amdgpu | tail -3; done
  ;;
al)
  # the lab switches added late in the round under test (fold limit 128, forced fold, LDS pad, plain-mode XCD mapping) + the whole GPU suite
  step 1100 pytest_r03al.log python -m pytest tests -m gpu -q
  grep -E "^FAILED|passed|failed" gpurun_out/pytest_r03al.log | tail -5
  ;;
am)
  # the 64-column shapes spill (k_spmm_mfma<., ., 64> at 256 VGPRs + 16-176 bytes of scratch): with and without the epilogue-operand prefetch (lab TFQMRGPU_EPI_PREFETCH)
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in st:32:64:c:64:64:4 st:64:64:c:45:45:4 st:16:64:c:90:90:4 st:32:64:z:45:45:4 st:64:64:z:32:32:4 st:16:64:z:64:64:4; do
    for pre in 1 0; do echo "$wl TFQMRGPU_EPI_PREFETCH=$pre"; TFQMRGPU_EPI_PREFETCH=$pre timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu; done
  done
  ;;
an)
  # 16 | 32 | 64 x 64 complex<float> on the quad-interleaved order, a wave per column half (k_spmm_ilvf, NH = 2): parity, then A/B against k_spmm_mfma (lab TFQMRGPU_ILV64=0)
  step 900 pytest_r03an.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py tests/test_gpu_mixed.py -q -x
  tail -4 gpurun_out/pytest_r03an.log
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in st:32:64:c:64:64:4 st:16:64:c:90:90:4 st:64:64:c:45:45:4; do
    for v in 0 1 0 1; do echo "$wl TFQMRGPU_ILV64=$v"; TFQMRGPU_ILV64=$v timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu; done
  done
  ;;
ao)
  # after the 64-column float kernel: the whole GPU suite
  step 1100 pytest_r03ao.log python -m pytest tests -m gpu -q
  grep -E "^FAILED|passed|failed" gpurun_out/pytest_r03ao.log | tail -5
  ;;
ap)
  # k_spmm_ilvz (16 | 32 | 64 x 32 | 64 complex<double>, row pairs interleaved): deviation report, parity, A/B against k_spmm_mfma (lab TFQMRGPU_ILVZ=0)
  step 300 zwide_report.txt python scripts/zwide_report.py
  grep -v amdgpu gpurun_out/zwide_report.txt | cut -c1-220
  step 900 pytest_r03ap.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py tests/test_gpu_configs.py -q
  grep -E "^FAILED|passed|failed" gpurun_out/pytest_r03ap.log | tail -8
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in st:32:32:z:64:64:4 st:16:32:z:90:90:4 st:16:64:z:64:64:4 st:32:64:z:45:45:4 st:64:64:z:32:32:4; do
    for v in 0 1 0 1; do echo "$wl TFQMRGPU_ILVZ=$v"; TFQMRGPU_ILVZ=$v timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu; done
  done
  ;;
aq)
  # after taking k_spmm_ilvz out again: the wide-z report on k_spmm_mfma, the whole GPU suite
  step 300 zwide_report.txt python scripts/zwide_report.py
  grep -v amdgpu gpurun_out/zwide_report.txt | cut -c1-220
  step 1100 pytest_r03aq.log python -m pytest tests -m gpu -q
  grep -E "^FAILED|passed|failed" gpurun_out/pytest_r03aq.log | tail -5
  ;;
ar)
  # mixed precision: the digits still missing split evenly over cycles of at most 3 digits (TFQMRGPU_MIXED_SPLIT=1) against a quarter of what is missing per cycle (=0)
  export TFQMRGPU_LIB=$PWD/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in fd2d_16x16_z stencil2d_8x8_z st:32:32:z:48:48:4 st:16:16:z:96:96:8 fd2d_16x16_z_small; do
    for v in 0 1 2; do echo "TFQMRGPU_MIXED_SPLIT=$v"; TFQMRGPU_MIXED_SPLIT=$v timeout 300 python scripts/mixed_trace.py $wl 2>&1 | grep -v amdgpu; done
  done
  TFQMRGPU_MIXED_SPLIT=0 timeout 300 python scripts/mixed_trace.py fd2d_16x16_z 1e-6 2>&1 | grep -v amdgpu
  TFQMRGPU_MIXED_SPLIT=1 timeout 300 python scripts/mixed_trace.py fd2d_16x16_z 1e-6 2>&1 | grep -v amdgpu
  TFQMRGPU_MIXED_SPLIT=1 timeout 300 python scripts/mixed_trace.py fd2d_16x16_z 1e-12 2>&1 | grep -v amdgpu
  TFQMRGPU_MIXED_SPLIT=0 timeout 300 python scripts/mixed_trace.py fd2d_16x16_z 1e-12 2>&1 | grep -v amdgpu
  unset TFQMRGPU_LIB
  step 600 pytest_r03ar.log python -m pytest tests/test_gpu_mixed.py tests/test_gpu_ranks.py -q
  tail -3 gpurun_out/pytest_r03ar.log
  ;;
at)
  # config 5 (8 x 8 z): timing-only probes of k_spmm_ilv8 -- 3 of 4 A fetches skipped (64), X fetches (128), both (192): is the L2 -> L1 operand path what bounds it?
  export AB_MAXIT=30
  timeout 800 python scripts/ab_fused.py stencil2d_8x8_z tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_p64.so scripts/bin/libtfQMRgpu_p128.so scripts/bin/libtfQMRgpu_p192.so tfqmrgpu_amd/lib/libtfQMRgpu.so 2>&1 | grep -v amdgpu
  ;;
au)
  # column batches (k_spmm_ilv8b: 8 x 8 z block columns with identical row patterns multiplied two at a time): parity, then A/B on config 5 (lab TFQMRGPU_BATCH=1: off)
  step 900 pytest_r03au.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py tests/test_gpu_configs.py tests/test_gpu_mixed.py tests/test_gpu_ranks.py -q -x
  tail -3 gpurun_out/pytest_r03au.log
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in stencil2d_8x8_z st:8:8:z:512:512:8 st:8:8:z:256:256:4; do
    for v in 1 2 1 2; do echo "$wl TFQMRGPU_BATCH=$v"; TFQMRGPU_BATCH=$v timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu; done
  done
  ;;
av)
  # column batches, epilogue operands requested in front of the products again: A/B on config 5 (lab TFQMRGPU_BATCH=1: off)
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in stencil2d_8x8_z st:8:8:z:512:512:8 st:8:8:z:256:256:4; do
    for v in 1 2 1 2; do echo "$wl TFQMRGPU_BATCH=$v"; TFQMRGPU_BATCH=$v timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu; done
  done
  ;;
aw)
  # column batches shipped: the whole GPU suite (with the new bit-identity and multiply tests), config 5 bench lines
  step 1100 pytest_r03aw.log python -m pytest tests -m gpu -q
  grep -E "^FAILED|passed|failed" gpurun_out/pytest_r03aw.log | tail -5
  python bench.py --workload stencil2d_8x8_z --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
  ;;
ax)
  # column batches of up to FOUR columns at two waves per SIMD (temporary build: kColBatchMax = 4, launch bound 2): lab TFQMRGPU_BATCH = 1 | 2 | 4 on config 5
  L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in stencil2d_8x8_z st:8:8:z:512:512:8; do
    for v in 1 2 4 1 2 4; do echo "$wl TFQMRGPU_BATCH=$v"; TFQMRGPU_BATCH=$v timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu; done
  done
  ;;
az)
  # 16 x 16 complex<float> (k_spmm_ilv16f; the shape and precision of the reference's bench multi default): the operand-skipping probes 64 | 128 | 192
  export AB_MAXIT=30
  timeout 800 python scripts/ab_fused.py st:16:16:c:96:96:16 tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_p64.so scripts/bin/libtfQMRgpu_p128.so scripts/bin/libtfQMRgpu_p192.so tfqmrgpu_amd/lib/libtfQMRgpu.so 2>&1 | grep -v amdgpu
  ;;
ba)
  # column batches: 2 | 3 | 4 block products in flight per wave (variant builds -DTFQ_B8_DEPTH)
  for wl in stencil2d_8x8_z st:8:8:z:512:512:8; do
    timeout 600 python scripts/ab_fused.py $wl tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_d3.so scripts/bin/libtfQMRgpu_d4.so tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_d3.so scripts/bin/libtfQMRgpu_d4.so 2>&1 | grep -v amdgpu
  done
  ;;
bb)
  # soak: 100 solves each must give the same bits -- config 5 (column pairs), P2 in 'm' (new cycle policy), a folded small system
  timeout 500 python scripts/soak.py stencil2d_8x8_z 100 2>&1 | grep -v amdgpu | tail -2
  timeout 500 python scripts/soak.py fd2d_16x16_z 60 m 2>&1 | grep -v amdgpu | tail -2
  timeout 300 python scripts/soak.py st:16:16:z:8:8:2 200 2>&1 | grep -v amdgpu | tail -2
  ;;
bd)
  # where the ~20 us of a small multiply go: stamps of one wave per work group of k_spmm_ilv16 (variant build -DTFQ_LAB_STAMPS), scripts/wg_timeline.py
  timeout 300 python scripts/wg_timeline.py 2>&1 | grep -v amdgpu
  ;;
be)
  # the same timeline under LOAD: P2, plain multiply and the last launches of the two fused multiplies of a solve
  for e in 0 1 2; do WG_EPI=$e timeout 300 python scripts/wg_timeline.py fd2d_16x16_z 2>&1 | grep -v amdgpu; done
  ;;
bf)
  # where requests queue: L1 -> L2 and L2 -> fabric read latencies per kernel from PMC (scripts/pmc_latency.sh)
  timeout 1000 bash scripts/pmc_latency.sh gpurun_out/pmc_lat > gpurun_out/r03_pmc_latency.txt 2>&1
  cat gpurun_out/r03_pmc_latency.txt | cut -c1-230
  ;;
*) echo "usage: r03.sh <step>; steps:"; grep -E "^[a-z]+\)$" "$0" | tr -d ")" | tr "\n" " "; echo; exit 2 ;;
esac
