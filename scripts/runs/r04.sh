#!/bin/bash
# Every GPU call of round 4 as one parametrised script: gpurun --timeout N -- bash scripts/runs/r04.sh <step>.  A step is what was one
# gpurun call; its comment says what it measured, profiles/r04_*.txt hold what came out (the records name the steps as "r04.sh <step>").
source scripts/gpu_steps.sh
# (steps a-e, j use kernels and switches that were measured and then taken out again -- k_spmm_ilv16p, k_spmm_ilv8c, scripts/pipe_bits.py: commit 34b750b has them;
#  steps ab, ac the k_spmm_small4 variant of commit cdceff4's successor: they document how the records under profiles/ were produced, they do not run on this tree)
case "$1" in
a)
  # round 4, first call: one wave per chunk with a pipeline across its Y blocks (k_spmm_ilv16p): bit-identity with k_spmm_ilv16, A/B on P2
  step 600 r04a_bits.txt python scripts/pipe_bits.py
  cat gpurun_out/r04a_bits.txt
  step 600 r04a_ab.txt python scripts/ab_fused.py fd2d_16x16_z lab@TFQMRGPU_PIPE=0 lab@TFQMRGPU_PIPE=1 lab@TFQMRGPU_PIPE=1,TFQMRGPU_PIPE_SORT=1 lab@TFQMRGPU_PIPE=0 default
  cat gpurun_out/r04a_ab.txt
  ;;
b)
  # one wave per chunk: is it the number of chunks in flight per XCD (L2 working set)?  unused LDS limits the work groups per CU: 100 KiB -> 1, 60 -> 2, 40 -> 3
  step 900 r04b_ab.txt python scripts/ab_fused.py fd2d_16x16_z lab@TFQMRGPU_PIPE=0 lab@TFQMRGPU_PIPE=1,TFQMRGPU_PIPE_LDS_KIB=100 lab@TFQMRGPU_PIPE=1,TFQMRGPU_PIPE_LDS_KIB=60 lab@TFQMRGPU_PIPE=1,TFQMRGPU_PIPE_LDS_KIB=40 lab@TFQMRGPU_PIPE=1
  cat gpurun_out/r04b_ab.txt
  ;;
c)
  # PMC of the two forms of the 16 x 16 z multiply on P2: one wave per Y block (PIPE=0) | one wave per chunk, pipelined (PIPE=1, two work groups per CU)
  export TFQMRGPU_LIB=$GRAFT_REPO_ROOT/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  export TFQMRGPU_PIPE=0
  bash scripts/pmc_collect.sh gpurun_out/r04c_pmc0 > gpurun_out/r04c_pmc0.log 2>&1
  python3 scripts/pmc_summary.py gpurun_out/r04c_pmc0 > gpurun_out/r04c_pmc0.json
  export TFQMRGPU_PIPE=1 TFQMRGPU_PIPE_LDS_KIB=60
  bash scripts/pmc_collect.sh gpurun_out/r04c_pmc1 > gpurun_out/r04c_pmc1.log 2>&1
  python3 scripts/pmc_summary.py gpurun_out/r04c_pmc1 > gpurun_out/r04c_pmc1.json
  rm -rf gpurun_out/r04c_pmc0 gpurun_out/r04c_pmc1
  tail -3 gpurun_out/r04c_pmc0.log gpurun_out/r04c_pmc1.log
  ;;
d)
  # one wave per chunk: 2 | 3 | 4 operand sets in flight per wave x work groups per CU (unused LDS: 100 KiB -> 1, 60 -> 2)
  step 900 r04d_ab.txt python scripts/ab_fused.py fd2d_16x16_z lab@TFQMRGPU_PIPE=0 lab@TFQMRGPU_PIPE_NSET=3 lab@TFQMRGPU_PIPE_NSET=3,TFQMRGPU_PIPE_LDS_KIB=100 lab@TFQMRGPU_PIPE_NSET=4 lab@TFQMRGPU_PIPE_NSET=4,TFQMRGPU_PIPE_LDS_KIB=60 lab@TFQMRGPU_PIPE_NSET=2,TFQMRGPU_PIPE_LDS_KIB=60
  cat gpurun_out/r04d_ab.txt
  ;;
e)
  # 8 x 8 z: column batches of four at work-group level, one wave per column (k_spmm_ilv8c): bit-identity, A/B on config 5
  step 900 r04e_bits.log python -m pytest tests/test_gpu_hash_mode.py -q -x -k "column_batches"
  step 900 r04e_ab.txt python scripts/ab_fused.py stencil2d_8x8_z lab@TFQMRGPU_BATCH_WIDE=0 lab@TFQMRGPU_BATCH_WIDE=1 lab@TFQMRGPU_BATCH=1 lab@TFQMRGPU_BATCH_WIDE=0 lab@TFQMRGPU_BATCH_WIDE=1
  cat gpurun_out/r04e_ab.txt
  ;;
f)
  # PMC of config 5 (8 x 8 z, 256^2 rows, 8 columns) with the shipped column pairs (k_spmm_ilv8b): what the multiply moves through the fabric
  export TFQMRGPU_LIB=$GRAFT_REPO_ROOT/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  export TFQMRGPU_BATCH_WIDE=0
  bash scripts/pmc_collect.sh gpurun_out/r04f_pmc stencil2d_8x8_z > gpurun_out/r04f_pmc.log 2>&1
  python3 scripts/pmc_summary.py gpurun_out/r04f_pmc > gpurun_out/r04f_pmc.json
  rm -rf gpurun_out/r04f_pmc
  tail -n 3 gpurun_out/r04f_pmc.log
  ;;
g)
  # k_v5_nrm's record sum with its LDS reads in batches (chunk_reduce), column sums with 24 | 12 records in flight: bits unchanged? times?
  step 900 r04g_tests.log python -m pytest tests/test_gpu_hash_mode.py tests/test_gpu_parity.py -q -x
  step 600 r04g_ab.txt python scripts/ab_fused.py fd2d_16x16_z scripts/bin/libtfQMRgpu_r03.so default scripts/bin/libtfQMRgpu_r03.so default
  cat gpurun_out/r04g_ab.txt
  ;;
h)
  # the same A/B with every kernel class listed (AB_ALL=1): the column kernels with 24 | 12 records in flight
  export AB_ALL=1
  step 600 r04h_ab.txt python scripts/ab_fused.py fd2d_16x16_z scripts/bin/libtfQMRgpu_r03.so default scripts/bin/libtfQMRgpu_r03.so default
  cat gpurun_out/r04h_ab.txt
  ;;
i)
  # column sums with 16 records in flight and a predicated tail (bits unchanged?), the Fortran example's three cases (A 'n', X 'n', B 't')
  export AB_ALL=1
  step 600 r04i_fortran.log python -m pytest tests/test_fortran.py -q -x
  cat gpurun_out/r04i_fortran.log | tail -n 20
  step 600 r04i_ab.txt python scripts/ab_fused.py fd2d_16x16_z scripts/bin/libtfQMRgpu_r03.so default scripts/bin/libtfQMRgpu_r03.so default
  cat gpurun_out/r04i_ab.txt
  step 900 r04i_tests.log python -m pytest tests/test_gpu_hash_mode.py tests/test_gpu_parity.py -q -x
  ;;
j)
  # the final form of k_spmm_ilv16p (counted loop, NSET operand sets) against k_spmm_ilv16, bit by bit
  step 900 r04j_bits.txt python scripts/pipe_bits.py
  cat gpurun_out/r04j_bits.txt
  ;;
k)
  # the product without the probe hooks, run_mixed's failure protocol, the fenced fold: the whole GPU suite
  step 1100 r04k_tests.log python -m pytest tests -q -x -m gpu
  ;;
l)
  # PMC passes of config 3 (32 x 32 c) and of 16 x 16 c with the product; the kernel-family getter; a bench line with the guarded traffic figure
  step 300 r04l_family.log python -m pytest tests/test_gpu_configs.py -q -x -k kernel_family
  bash scripts/pmc_collect.sh gpurun_out/r04l_pmc3 stencil3d_32x32_c > gpurun_out/r04l_pmc3.log 2>&1
  python3 scripts/pmc_summary.py gpurun_out/r04l_pmc3 > gpurun_out/r04l_pmc_32x32c.json
  bash scripts/pmc_collect.sh gpurun_out/r04l_pmc16c st:16:16:c:96:96:16 > gpurun_out/r04l_pmc16c.log 2>&1
  python3 scripts/pmc_summary.py gpurun_out/r04l_pmc16c > gpurun_out/r04l_pmc_16x16c.json
  rm -rf gpurun_out/r04l_pmc3 gpurun_out/r04l_pmc16c
  step 500 r04l_bench.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline
  tail -c 1500 gpurun_out/r04l_bench.json
  ;;
m)
  # the price of the stopping test's collectives on one rank (bench.py: collectives): P2 and config 3
  step 600 r04m_bench_p2.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mixed --no-hbm-multiply
  step 600 r04m_bench_cfg3.json python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload stencil3d_32x32_c
  python3 -c "
import json
for f in ('gpurun_out/r04m_bench_p2.json', 'gpurun_out/r04m_bench_cfg3.json'):
    p = json.loads(open(f).read().strip().splitlines()[-1]); print(f, p['ms_per_step'], json.dumps(p['collectives']))
"
  ;;
n)
  # the native-order multiply with a prepared launch order (tfqmrgpuExt_multiplyPrepare): config 1's plan file, P2's native listing
  step 600 r04n_native.txt python scripts/native_order_probe.py 50
  cat gpurun_out/r04n_native.txt
  ;;
o)
  # ... and with 3 | 4 operand sets (block products in flight per wave) in the 16 x 16 plain-mode kernel (lab switch TFQMRGPU_PLAIN_NSET)
  export TFQMRGPU_LIB=$GRAFT_REPO_ROOT/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for n in 2 3 4; do
    export TFQMRGPU_PLAIN_NSET=$n
    echo "## TFQMRGPU_PLAIN_NSET=$n" >> gpurun_out/r04o_native.txt
    step 600 r04o_native_$n.txt python scripts/native_order_probe.py 50
    grep -v "mode 2" gpurun_out/r04o_native_$n.txt | sort -u >> gpurun_out/r04o_native.txt
  done
  cat gpurun_out/r04o_native.txt
  ;;
p)
  # prepared order, the XCDs splitting the rows (mode 3) or the library choosing (4)
  step 600 r04p_native.txt python scripts/native_order_probe.py 50
  cat gpurun_out/r04p_native.txt
  ;;
q)
  # prepared order in the tests and in both bench drivers (config 1: bench_tfqmrgpu multi)
  step 900 r04q_tests.log python -m pytest tests/test_gpu_parity.py tests/test_bench_binary.py -q -x -k "multiply or bench or binary"
  for p in f z; do
    BENCH_ORDER=0 tfqmrgpu_amd/lib/bench_tfqmrgpu multi tests/golden/plan_unordered.14-287-16.gz $p 200 3 > gpurun_out/r04q_multi_${p}_order0.txt 2>&1
    tfqmrgpu_amd/lib/bench_tfqmrgpu multi tests/golden/plan_unordered.14-287-16.gz $p 200 3 > gpurun_out/r04q_multi_${p}_order4.txt 2>&1
    grep -h "launch order\|GPU performance\|roofline" gpurun_out/r04q_multi_${p}_order0.txt gpurun_out/r04q_multi_${p}_order4.txt
  done
  ;;
r)
  # where the iteration of the small block shapes goes: every kernel class (AB_ALL=1), 5-point stencils of about 256 MB per vector, 4 block columns
  export AB_ALL=1
  for wl in st:4:4:z:512:512:4 st:4:5:z:457:457:4 st:4:8:z:362:362:4 st:8:9:z:241:241:4 st:8:10:z:228:228:4 st:16:16:z:128:128:4 st:4:4:c:724:724:4 st:4:5:c:647:647:4; do
    echo "## $wl" >> gpurun_out/r04r_shapes.txt
    step 300 r04r_one.txt python scripts/ab_fused.py $wl default
    grep -v amdgpu.ids gpurun_out/r04r_one.txt >> gpurun_out/r04r_shapes.txt
  done
  cat gpurun_out/r04r_shapes.txt
  ;;
s)
  # columns of more than 256 chunk records summed by several work groups (column_total): tests, then the column kernels on P2, configs 3 and 5, a 4-column stencil
  step 1100 r04s_tests.log python -m pytest tests -q -x -m gpu
  export AB_ALL=1
  for wl in fd2d_16x16_z stencil3d_32x32_c stencil2d_8x8_z st:16:16:z:128:128:4; do
    echo "## $wl" >> gpurun_out/r04s_ab.txt
    step 400 r04s_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_r03.so default
    grep -v amdgpu.ids gpurun_out/r04s_one.txt >> gpurun_out/r04s_ab.txt
  done
  cat gpurun_out/r04s_ab.txt
  ;;
t)
  # columns of more than 1024 chunk records summed by several work groups: previous commit | this one, every kernel class
  export AB_ALL=1
  for wl in fd2d_16x16_z stencil3d_32x32_c stencil2d_8x8_z st:16:16:z:128:128:4 st:16:16:z:128:128:32; do
    echo "## $wl" >> gpurun_out/r04t_ab.txt
    step 400 r04t_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04t_one.txt >> gpurun_out/r04t_ab.txt
  done
  cat gpurun_out/r04t_ab.txt
  step 600 r04t_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_ranks.py -q -x
  ;;
u)
  # the same with the shares as memory-side atomic exchanges (no fence): previous commit | this one
  export AB_ALL=1
  step 600 r04u_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_ranks.py -q -x
  for wl in stencil3d_32x32_c stencil2d_8x8_z st:16:16:z:128:128:4 st:16:16:z:128:128:32; do
    echo "## $wl" >> gpurun_out/r04u_ab.txt
    step 400 r04u_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04u_one.txt >> gpurun_out/r04u_ab.txt
  done
  cat gpurun_out/r04u_ab.txt
  ;;
v)
  # segments of one round of loads (256 / LN x 16 records), columns of more than four rounds cut: whole GPU suite, then previous commit | this one
  step 1100 r04v_tests.log python -m pytest tests -q -x -m gpu
  export AB_ALL=1
  for wl in fd2d_16x16_z stencil3d_32x32_c stencil2d_8x8_z st:16:16:z:128:128:4 st:16:16:z:128:128:32; do
    echo "## $wl" >> gpurun_out/r04v_ab.txt
    step 400 r04v_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04v_one.txt >> gpurun_out/r04v_ab.txt
  done
  cat gpurun_out/r04v_ab.txt
  ;;
w)
  # folding without fences (co_store / co_load, relaxed arrival): the fold on / off bit-identity tests, then the crossover (never | always fold) by chunk count
  step 900 r04w_tests.log python -m pytest tests/test_gpu_hash_mode.py -q -x -k "switches or previous_buffer or fold"
  tail -n 3 gpurun_out/r04w_tests.log
  step 900 r04w_crossover.txt python scripts/fold_crossover.py "FD:1.75,6.75,2,3,0.0,4" "FD:6,24,4,2,-0.25,4" st:16:16:z:8:8:4 st:16:16:z:16:16:4 st:16:16:z:24:24:4 st:16:16:z:32:32:4 st:16:16:z:48:48:4 st:16:16:z:64:64:4 st:16:16:z:96:96:4 st:16:16:z:128:128:4 st:16:16:c:48:48:4 st:32:32:c:24:24:2 stencil3d_32x32_c st:8:8:z:64:64:4 fd2d_16x16_z
  grep -v amdgpu.ids gpurun_out/r04w_crossover.txt
  ;;
x)
  # fold limit 384 chunks: the whole GPU suite
  step 1100 r04x_tests.log python -m pytest tests -q -x -m gpu
  tail -n 4 gpurun_out/r04x_tests.log
  ;;
y)
  # mixed precision: the plan remembers its float floor, a short last cycle runs a predicted number of iterations without probes
  step 900 r04y_tests.log python -m pytest tests/test_gpu_mixed.py tests/test_gpu_ranks.py -q -x
  tail -n 3 gpurun_out/r04y_tests.log
  export TFQMRGPU_LIB=$GRAFT_REPO_ROOT/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
  for wl in fd2d_16x16_z stencil2d_8x8_z st:32:32:z:48:48:4 st:16:16:z:96:96:8 fd2d_16x16_z_small; do
    for sw in "0 0" "1 0" "0 1" "1 1"; do
      set -- $sw
      echo "TFQMRGPU_MIXED_FLOOR=$1 TFQMRGPU_MIXED_PREDICT=$2" >> gpurun_out/r04y_mixed.txt
      TFQMRGPU_MIXED_FLOOR=$1 TFQMRGPU_MIXED_PREDICT=$2 python scripts/mixed_trace.py $wl 2>&1 | grep -v amdgpu.ids | grep " m status" >> gpurun_out/r04y_mixed.txt
    done
  done
  cat gpurun_out/r04y_mixed.txt
  ;;
z)
  # mixed precision, first cycle asked for twice the remembered floor: P2 and the others again, product build
  step 900 r04z_tests.log python -m pytest tests/test_gpu_mixed.py -q -x
  for wl in fd2d_16x16_z stencil2d_8x8_z st:32:32:z:48:48:4 st:16:16:z:96:96:8 fd2d_16x16_z_small stencil3d_32x32_c; do
    python scripts/mixed_trace.py $wl 2>&1 | grep -v amdgpu.ids | grep " m status" >> gpurun_out/r04z_mixed.txt
  done
  cat gpurun_out/r04z_mixed.txt
  ;;
aa)
  # the mixed-precision tests with the remembered floor
  step 900 r04aa_tests.log python -m pytest tests/test_gpu_mixed.py -q -x
  tail -n 5 gpurun_out/r04aa_tests.log
  ;;
ab)
  # k_spmm_small4 with 16-byte operand pieces and four block products in flight: all-shapes tests, then previous commit | this one on the 4 x N systems
  step 900 r04ab_tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py -q -x -k "block_sizes or 4x4 or fd_4"
  tail -n 3 gpurun_out/r04ab_tests.log
  for wl in st:4:4:z:512:512:4 st:4:5:z:457:457:4 st:4:8:z:362:362:4 st:4:4:c:724:724:4 st:4:5:c:647:647:4 st:4:8:c:512:512:4; do
    echo "## $wl" >> gpurun_out/r04ab_ab.txt
    step 400 r04ab_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04ab_one.txt >> gpurun_out/r04ab_ab.txt
  done
  cat gpurun_out/r04ab_ab.txt
  ;;
ac)
  # k_spmm_small4: 16-byte pieces with 1 | 2 | 4 products in flight against the previous commit
  for wl in st:4:4:z:512:512:4 st:4:5:z:457:457:4 st:4:4:c:724:724:4; do
    echo "## $wl" >> gpurun_out/r04ac_ab.txt
    step 400 r04ac_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so scripts/bin/libtfQMRgpu_s4d1.so scripts/bin/libtfQMRgpu_s4d2.so default
    grep -v amdgpu.ids gpurun_out/r04ac_one.txt >> gpurun_out/r04ac_ab.txt
  done
  cat gpurun_out/r04ac_ab.txt
  ;;
ad)
  # what a device-wide barrier inside one cooperative launch costs against a boundary between dependent launches (scripts/grid_sync_probe.hip)
  step 120 r04ad_grid_sync.txt scripts/bin/grid_sync_probe
  cat gpurun_out/r04ad_grid_sync.txt
  ;;
ae)
  # the timeline of one P2 solve: kernel durations and the gaps between them (rocprofv3 --kernel-trace, scripts/trace_gaps.py)
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  rm -rf gpurun_out/r04ae_trace
  step 300 r04ae_trace.log rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04ae_trace -- python3 scripts/small_one.py 16 120 4 2
  python3 scripts/trace_gaps.py gpurun_out/r04ae_trace 400 > gpurun_out/r04ae_gaps.txt
  rm -rf gpurun_out/r04ae_trace
  python3 - <<'PY'
import re
tot = gap = 0.0; n = 0; big = []
for l in open("gpurun_out/r04ae_gaps.txt"):
    m = re.match(r"\s*([\d.]+) us  gap\s+(-?[\d.]+)  dur\s+([\d.]+)  (.*)", l)
    if not m: continue
    n += 1; g = float(m.group(2)); d = float(m.group(3)); tot += d; gap += max(0.0, g)
    if g > 6: big.append((float(m.group(1)), g, m.group(4)))
    last = float(m.group(1)) + d
print("kernels %d, sum of durations %.1f us, sum of gaps %.1f us, span %.1f us" % (n, tot, gap, last))
for b in big[:30]: print("gap %.1f us at %.1f us before %s" % (b[1], b[0], b[2]))
PY
  ;;
af)
  # k_x_v6_v7 with two items per trip, loads of both in front: previous commit | this one
  for wl in fd2d_16x16_z stencil2d_8x8_z stencil3d_32x32_c; do
    echo "## $wl" >> gpurun_out/r04af_ab.txt
    step 400 r04af_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04af_one.txt >> gpurun_out/r04af_ab.txt
  done
  cat gpurun_out/r04af_ab.txt
  step 600 r04af_tests.log python -m pytest tests/test_gpu_hash_mode.py tests/test_gpu_parity.py -q -x
  tail -n 3 gpurun_out/r04af_tests.log
  ;;
ag)
  # 8 x 9 and 8 x 10 z on the row-pair-interleaved order (k_spmm_ilv8w with a ragged second column group): the whole GPU suite, then lab ILV89 = 0 | 1
  step 1100 r04ag_tests.log python -m pytest tests -q -x -m gpu
  tail -n 3 gpurun_out/r04ag_tests.log
  export AB_ALL=1
  for wl in st:8:9:z:241:241:4 st:8:10:z:228:228:4; do
    echo "## $wl" >> gpurun_out/r04ag_ab.txt
    step 400 r04ag_one.txt python scripts/ab_fused.py $wl lab@TFQMRGPU_ILV89=0 lab@TFQMRGPU_ILV89=1 default
    grep -v amdgpu.ids gpurun_out/r04ag_one.txt >> gpurun_out/r04ag_ab.txt
  done
  cat gpurun_out/r04ag_ab.txt
  ;;
ah)
  # soak: the fence-free hand-overs (folded column operations, segment shares of long columns) must never change a bit -- N solves per workload, one status / iteration count / residual / md5 of X
  for spec in "fd2d_16x16_z_small 400" "st:16:16:z:16:16:4 400" "st:8:8:z:24:24:2 400" "stencil3d_32x32_c 300" "stencil2d_8x8_z 60" "st:16:16:z:128:128:4 60" "fd2d_16x16_z 30" "fd2d_16x16_z_small 200 m" "st:8:9:z:60:60:3 200"; do
    set -- $spec
    step 500 r04ah_one.txt python scripts/soak.py "$@"
    tail -n 1 gpurun_out/r04ah_one.txt >> gpurun_out/r04ah_soak.txt
  done
  cat gpurun_out/r04ah_soak.txt
  ;;
ai)
  # all 15 block shapes x z, c at the end of the round (scripts/shape_survey.sh)
  step 1100 r04ai_shape_survey.txt bash scripts/shape_survey.sh
  grep -v amdgpu.ids gpurun_out/r04ai_shape_survey.txt | cut -c1-260
  ;;
aj)
  # launch order along the strongest band coupling (lab: TFQMRGPU_ORDER=2) x column-group size, on config 5, P2, config 3 and the config-4 shard
  for wl in stencil2d_8x8_z fd2d_16x16_z stencil3d_32x32_c st:16:16:z:128:128:32; do
    echo "## $wl" >> gpurun_out/r04aj_ab.txt
    step 600 r04aj_one.txt python scripts/ab_fused.py $wl lab@TFQMRGPU_ORDER=1 lab@TFQMRGPU_ORDER=2 lab@TFQMRGPU_ORDER=2,TFQMRGPU_ORDER_G=8 lab@TFQMRGPU_ORDER=2,TFQMRGPU_ORDER_G=16
    grep -v amdgpu.ids gpurun_out/r04aj_one.txt >> gpurun_out/r04aj_ab.txt
  done
  cat gpurun_out/r04aj_ab.txt
  ;;
ak)
  # column operations with their per-RHS scalars requested in front of the column's sum: previous commit | this one, every class; then the tests that hold the bits
  export AB_ALL=1
  for wl in stencil3d_32x32_c fd2d_16x16_z "FD:6,24,4,2,-0.25,4"; do
    echo "## $wl" >> gpurun_out/r04ak_ab.txt
    if [ "${wl#FD:}" != "$wl" ]; then
      for lib in scripts/bin/libtfQMRgpu_prev.so tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_prev.so tfqmrgpu_amd/lib/libtfQMRgpu.so; do
        TFQMRGPU_LIB=$GRAFT_REPO_ROOT/$lib python scripts/fold_crossover.py "$wl" 2>&1 | grep -v amdgpu.ids | grep "fold_max 0 " | sed -e "s|^|$lib |" >> gpurun_out/r04ak_ab.txt
      done
    else
      step 400 r04ak_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default scripts/bin/libtfQMRgpu_prev.so default
      grep -v amdgpu.ids gpurun_out/r04ak_one.txt >> gpurun_out/r04ak_ab.txt
    fi
  done
  cat gpurun_out/r04ak_ab.txt
  step 900 r04ak_tests.log python -m pytest tests/test_gpu_hash_mode.py tests/test_gpu_parity.py tests/test_gpu_mixed.py -q -x
  tail -n 3 gpurun_out/r04ak_tests.log
  ;;
al)
  # k_spmm_small4 with three 20-lane thread groups per wave for 4 x 5 blocks (two 32-lane groups before): previous commit | this one, z and c; 4 x 4 as the control
  export AB_ALL=1
  for wl in st:4:5:z:457:457:4 st:4:5:c:647:647:4 st:4:4:z:512:512:4; do
    echo "## $wl" >> gpurun_out/r04al_ab.txt
    step 400 r04al_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04al_one.txt >> gpurun_out/r04al_ab.txt
  done
  cat gpurun_out/r04al_ab.txt
  step 900 r04al_tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_operator.py tests/test_gpu_mixed.py -q -x
  tail -n 3 gpurun_out/r04al_tests.log
  ;;
am)
  # k_spmm_m4 (v_mfma_f64_4x4x4_4b_f64, chunk index data in LDS) against k_spmm_small4 on 4 x 4 and 4 x 8 z: lab switch TFQMRGPU_M4 = 0 | 1 on one build
  step 900 r04am_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_operator.py -q -x
  tail -n 3 gpurun_out/r04am_tests.log
  export AB_ALL=1
  for wl in st:4:4:z:512:512:4 st:4:8:z:362:362:4; do
    echo "## $wl" >> gpurun_out/r04am_ab.txt
    step 400 r04am_one.txt python scripts/ab_fused.py $wl lab@TFQMRGPU_M4=0 lab@TFQMRGPU_M4=1 lab@TFQMRGPU_M4=0 lab@TFQMRGPU_M4=1
    grep -v amdgpu.ids gpurun_out/r04am_one.txt >> gpurun_out/r04am_ab.txt
  done
  cat gpurun_out/r04am_ab.txt
  ;;
an)
  # k_spmm_m4: products requested per trip (TFQ_M4_NB = 4 | 5 | 6 | 8 = default), 5-point and 13-point stencils
  export AB_ALL=1
  for wl in st:4:4:z:512:512:4 st:4:8:z:362:362:4 st:4:4:z:512:512:4:13; do
    echo "## $wl" >> gpurun_out/r04an_ab.txt
    step 400 r04an_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_nb4.so scripts/bin/libtfQMRgpu_nb5.so scripts/bin/libtfQMRgpu_nb6.so default
    grep -v amdgpu.ids gpurun_out/r04an_one.txt | grep -v "^    " >> gpurun_out/r04an_ab.txt
  done
  cat gpurun_out/r04an_ab.txt
  ;;
ap)
  # k_spmm_m4 with two neighbouring columns per lane (16-byte X and epilogue accesses) where LN % 8 == 0: 4 x 8 and 4 x 32 z, lab switch TFQMRGPU_M4 = 0 | 1 (0 = small4 | the tile kernel)
  step 900 r04ap_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_operator.py tests/test_gpu_mixed.py -q
  tail -n 3 gpurun_out/r04ap_tests.log
  export AB_ALL=1
  for wl in st:4:8:z:362:362:4 st:4:32:z:181:181:4 st:4:4:z:512:512:4; do
    echo "## $wl" >> gpurun_out/r04ap_ab.txt
    step 400 r04ap_one.txt python scripts/ab_fused.py $wl lab@TFQMRGPU_M4=0 lab@TFQMRGPU_M4=1 lab@TFQMRGPU_M4=0 lab@TFQMRGPU_M4=1
    grep -v amdgpu.ids gpurun_out/r04ap_one.txt | grep -v "^    " >> gpurun_out/r04ap_ab.txt
  done
  cat gpurun_out/r04ap_ab.txt
  ;;
aq)
  # A blocks of the k_spmm_m4 plans as complex pairs [k][i][re, im] (one 16-byte A access per lane and product): lab switch TFQMRGPU_A_CPLX = 0 | 1 (slower, not shipped: the change is not in the tree)
  step 900 r04aq_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_operator.py tests/test_gpu_mixed.py -q
  tail -n 3 gpurun_out/r04aq_tests.log
  export AB_ALL=1
  for wl in st:4:4:z:512:512:4 st:4:8:z:362:362:4 st:4:32:z:181:181:4; do
    echo "## $wl" >> gpurun_out/r04aq_ab.txt
    step 400 r04aq_one.txt python scripts/ab_fused.py $wl lab@TFQMRGPU_A_CPLX=0 lab@TFQMRGPU_A_CPLX=1 lab@TFQMRGPU_A_CPLX=0 lab@TFQMRGPU_A_CPLX=1
    grep -v amdgpu.ids gpurun_out/r04aq_one.txt | grep -v "^    " >> gpurun_out/r04aq_ab.txt
  done
  cat gpurun_out/r04aq_ab.txt
  ;;
ar)
  # k_spmm_small4 with its chunk's index data in LDS and the operands of up to 6 | 8 products requested at once: previous commit | this one
  step 900 r04ar_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_operator.py tests/test_gpu_mixed.py -q
  tail -n 3 gpurun_out/r04ar_tests.log
  export AB_ALL=1
  for wl in st:4:5:z:457:457:4 st:4:4:c:724:724:4 st:4:5:c:647:647:4 st:4:8:c:512:512:4 st:4:32:c:256:256:4; do
    echo "## $wl" >> gpurun_out/r04ar_ab.txt
    step 400 r04ar_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04ar_one.txt | grep -v "^    " >> gpurun_out/r04ar_ab.txt
  done
  cat gpurun_out/r04ar_ab.txt
  ;;
as)
  # k_spmm_ilv8w with the operands of 4 | 6 (8 x 9, 8 x 10) and 3 | 4 (8 x 32, 8 x 64) products requested at once instead of two refilled sets: variant builds w4 | w6 against the tree (slower: the change is not in the tree)
  export AB_ALL=1
  for wl in st:8:9:z:241:241:4 st:8:10:z:228:228:4 st:8:32:z:128:128:4 st:8:64:z:90:90:4; do
    echo "## $wl" >> gpurun_out/r04as_ab.txt
    step 400 r04as_one.txt python scripts/ab_fused.py $wl default scripts/bin/libtfQMRgpu_w4.so scripts/bin/libtfQMRgpu_w6.so default
    grep -v amdgpu.ids gpurun_out/r04as_one.txt | grep -v "^    " >> gpurun_out/r04as_ab.txt
  done
  cat gpurun_out/r04as_ab.txt
  ;;
at)
  # k_spmm_mfma8 (8 x 9 | 10 c) with its row ranges and index pairs as scalar loads: previous commit | tree
  step 900 r04at_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_operator.py tests/test_gpu_mixed.py -q
  tail -n 3 gpurun_out/r04at_tests.log
  export AB_ALL=1
  for wl in st:8:9:c:341:341:4 st:8:10:c:323:323:4; do
    echo "## $wl" >> gpurun_out/r04at_ab.txt
    step 400 r04at_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04at_one.txt | grep -v "^    " >> gpurun_out/r04at_ab.txt
  done
  cat gpurun_out/r04at_ab.txt
  ;;
au)
  # the shadow key's index reads (block row of the Y block, original column) as scalar loads in the hash-mode kernels: previous commit | tree, P2, config 5, 16 x 16 c  (1-3 % slower on the norm multiply: not in the tree)
  export AB_ALL=1
  for wl in fd2d_16x16_z stencil2d_8x8_z st:16:16:c:181:181:4; do
    echo "## $wl" >> gpurun_out/r04au_ab.txt
    step 500 r04au_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04au_one.txt | grep -v "^    " >> gpurun_out/r04au_ab.txt
  done
  cat gpurun_out/r04au_ab.txt
  step 900 r04au_tests.log python -m pytest tests/test_gpu_hash_mode.py tests/test_gpu_parity.py -q
  tail -n 3 gpurun_out/r04au_tests.log
  ;;
aw)
  # pageable host arrays through two pinned bounce buffers (tfq_api.hip: piped_copy) against one hipMemcpy: lab switch TFQMRGPU_PIPED_COPY = 0 | 1 (slower; the pipeline is not in the tree)
  step 600 r04aw_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -q
  tail -n 3 gpurun_out/r04aw_tests.log
  for wl in fd2d_16x16_z stencil2d_8x8_z; do
    for v in 0 1 0 1; do
      TFQMRGPU_LIB=$GRAFT_REPO_ROOT/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so TFQMRGPU_PIPED_COPY=$v python scripts/host_arrays.py $wl 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04aw_host_arrays.txt
    done
  done
  cat gpurun_out/r04aw_host_arrays.txt
  ;;
ay)
  # k_spmm_s4w (float 4 x 4 | 8 | 32: four neighbouring columns per lane, 16-byte accesses) against k_spmm_small4: lab switch TFQMRGPU_S4W = 0 | 1
  step 900 r04ay_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_operator.py tests/test_gpu_mixed.py -q
  tail -n 3 gpurun_out/r04ay_tests.log
  export AB_ALL=1
  for wl in st:4:4:c:724:724:4 st:4:8:c:512:512:4 st:4:32:c:256:256:4; do
    echo "## $wl" >> gpurun_out/r04ay_ab.txt
    step 400 r04ay_one.txt python scripts/ab_fused.py $wl lab@TFQMRGPU_S4W=0 lab@TFQMRGPU_S4W=1 lab@TFQMRGPU_S4W=0 lab@TFQMRGPU_S4W=1
    grep -v amdgpu.ids gpurun_out/r04ay_one.txt | grep -v "^    " >> gpurun_out/r04ay_ab.txt
  done
  cat gpurun_out/r04ay_ab.txt
  ;;
bh)
  # vector kernels with cached instead of non-temporal accesses where a block plane is not a whole number of 128-byte lines: previous commit | tree  (net slower on 5 of 6 shapes: not in the tree)
  step 900 r04bh_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_hash_mode.py -q
  tail -n 3 gpurun_out/r04bh_tests.log
  export AB_ALL=1
  for wl in st:4:5:z:457:457:4 st:8:9:z:241:241:4 st:4:4:c:724:724:4 st:4:5:c:647:647:4 st:8:9:c:341:341:4 st:8:10:c:323:323:4 st:4:4:z:512:512:4; do
    echo "## $wl" >> gpurun_out/r04bh_ab.txt
    step 400 r04bh_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04bh_one.txt | grep -v "^    " >> gpurun_out/r04bh_ab.txt
  done
  cat gpurun_out/r04bh_ab.txt
  ;;
bi)
  # soak of the new 4-row kernels (k_spmm_m4, k_spmm_small4 with its index data in LDS, k_spmm_s4w, k_spmm_mfma8 with scalar index loads): small (folded) and medium plans, z | c | m
  for spec in "st:4:4:z:24:24:3 300" "st:4:8:z:64:64:4 200" "st:4:32:z:40:40:2 200" "st:4:5:z:64:64:3 200" "st:4:4:c:64:64:4 200" "st:4:32:c:40:40:2 200" "st:8:9:c:40:40:2 200" "st:4:8:z:24:24:3 200 m" "st:4:4:z:256:256:4 40"; do
    set -- $spec
    step 500 r04bi_one.txt python scripts/soak.py "$@"
    tail -n 1 gpurun_out/r04bi_one.txt >> gpurun_out/r04bi_soak.txt
  done
  cat gpurun_out/r04bi_soak.txt
  ;;
bj)
  # k_spmm_m4 for LN = 4: neighbouring lanes share their memory instructions (even lane: the Re plane's two elements, odd lane: the Im plane's, one exchange): previous commit | tree  (slower: not in the tree)
  step 900 r04bj_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_operator.py tests/test_gpu_mixed.py tests/test_gpu_hash_mode.py -q
  tail -n 3 gpurun_out/r04bj_tests.log
  export AB_ALL=1
  for wl in st:4:4:z:512:512:4 st:4:4:z:512:512:4:13 st:4:4:z:24:24:3; do
    echo "## $wl" >> gpurun_out/r04bj_ab.txt
    step 400 r04bj_one.txt python scripts/ab_fused.py $wl scripts/bin/libtfQMRgpu_prev.so default scripts/bin/libtfQMRgpu_prev.so default
    grep -v amdgpu.ids gpurun_out/r04bj_one.txt | grep -v "^    " >> gpurun_out/r04bj_ab.txt
  done
  cat gpurun_out/r04bj_ab.txt
  ;;
br)
  # 8 x 9 and 8 x 10 c on the quad-interleaved order (k_spmm_ilv8f with a ragged second column group) against k_spmm_mfma8 on the native order: lab switch TFQMRGPU_ILV89F = 0 | 1  (fused multiplies slower: not in the tree)
  step 900 r04br_tests.log python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_operator.py tests/test_gpu_mixed.py tests/test_gpu_hash_mode.py -q
  tail -n 3 gpurun_out/r04br_tests.log
  export AB_ALL=1
  for wl in st:8:9:c:341:341:4 st:8:10:c:323:323:4; do
    echo "## $wl" >> gpurun_out/r04br_ab.txt
    step 400 r04br_one.txt python scripts/ab_fused.py $wl lab@TFQMRGPU_ILV89F=0 lab@TFQMRGPU_ILV89F=1 lab@TFQMRGPU_ILV89F=0 lab@TFQMRGPU_ILV89F=1
    grep -v amdgpu.ids gpurun_out/r04br_one.txt | grep -v "^    " >> gpurun_out/r04br_ab.txt
  done
  cat gpurun_out/r04br_ab.txt
  ;;
*) echo "unknown step $1"; exit 1;;
esac
