#!/usr/bin/env python3
"""Benchmark of the hot path: tfQMR solves of a block-sparse system through the C-ABI of libtfQMRgpu.so.

Workload (BASELINE.json configs[1]): the finite-difference example `generate_FD_example 16 120 4 2 -0.25`
(2-D, block edge 4 -> 16x16 complex<double> blocks; mb=3573, nnzbA=17589, nnzbX=138229, 49 block
columns = 784 right-hand sides, 679189 block products per multiply), solved to 1e-9.
A "step" is one complete solve (setup, all iterations with both multiplies, residual probes); the
matrices are resident in HBM before the timed region.  `value` = reference flop count of the solves
(tfqmrgpu_bsrsv_getInfo, the number the reference's own `bench_tfqmrgpu tfQMR` divides by its solver
time, bench_tfqmrgpu.cu:200-204) / wall time, summed over all GPUs.

--gpus N, one process per GPU: weak scaling by default, every rank owns its own 49 block columns of a N*49-column
system (same A, same sparsity, own shadow vector); `--workload cfg4` is BASELINE config 4, strong scaling: the 256
block columns of one 16x16 complex<double> system split N ways with the product's tfqmrgpuExt_shardColumns.  The only
communication is the RCCL max-all-reduce of the stopping-test scalars inside the solver.
Launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` (what the driver does), or
plain `python bench.py --gpus N`: bench.py then starts that launcher itself as a CHILD process before anything in this
process has touched the GPU, relays rank 0's JSON line and exits with the child's code (`--launcher` forces the same
path for N = 1: process group + RCCL communicator with one rank).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}  # dense matrix peaks (f64: SURVEY.md 8d public spec, f32: guide)


def build_problem(name, rank, world=1):
    from tfqmrgpu_amd import problems as PR
    from tfqmrgpu_amd.fd_generator import FDExample
    if name == "fd2d_16x16_z":
        ex = FDExample(16, 120, 4, 2, -0.25, 4)
        pr, prec, desc = ex.problem(), "z", "generate_FD_example 16 120 4 2 -0.25 n 0 4 (16x16 complex<double>)"
    elif name == "fd2d_16x16_z_small":
        ex = FDExample(6, 24, 4, 2, -0.25, 4)
        pr, prec, desc = ex.problem(), "z", "generate_FD_example 6 24 4 2 -0.25 n 0 4 (16x16 complex<double>)"
    elif name == "stencil3d_32x32_c":   # BASELINE configs[2]: 32x32 complex<float>, ~50k A blocks, 64 RHS
        pr = PR.stencil_2d(64, 64, 32, 32, 2, seed=3, points=13)
        pr.tolerance = 1e-4
        prec, desc = "c", "13-point block stencil 64x64, 32x32 complex<float>, 2 block columns (64 RHS)"
    elif name == "stencil2d_8x8_z":     # BASELINE configs[4]: 8x8 complex<double>, ~5 nnz/row
        pr = PR.stencil_2d(256, 256, 8, 8, 8, seed=5)
        prec, desc = "z", "5-point block stencil 256x256, 8x8 complex<double>, 8 block columns"
    elif name.startswith("st:"):        # st:LM:LN:prec:nx:ny:ncols  (5-point block stencil, X dense in ncols block columns)
        _, lm, ln, prec, nx, ny, nc = name.split(":")[:7]
        points = int(name.split(":")[7]) if name.count(":") > 6 else 5      # st:...:13 = the 13-point stencil
        pr = PR.stencil_2d(int(nx), int(ny), int(lm), int(ln), int(nc), seed=7, points=points)
        if prec == "c":
            pr.tolerance = 1e-4
        desc = "5-point block stencil %sx%s, %sx%s complex<%s>, %s block columns" % (nx, ny, lm, ln, "double" if prec == "z" else "float", nc)
    elif name == "cfg4" or name.startswith("cfg4:"):   # BASELINE configs[3]: 256 block columns (4096 RHS) of ONE system, sharded over the ranks
        # cfg4:nx:ncols = the same at a reduced size (tests).  What is built for all columns on every rank is index lists only (the
        # pattern of X, 4 bytes per block; tfqmrgpuExt_shardColumns cuts it); values exist for A (the same on every rank) and for the
        # shard's own B blocks -- no rank ever holds X-shaped values of another rank's columns
        import tfqmrgpu_amd as T
        nx, ncols = (int(v) for v in name.split(":")[1:3]) if ":" in name else (128, 256)
        full = PR.stencil_2d(nx, nx, 16, 16, ncols, seed=4)       # index lists of X (no values), A, one B block per column
        pr, _, _ = T.shard_columns(full, world, rank)
        desc = ("5-point block stencil %dx%d, 16x16 complex<double>, %d block columns (%d RHS) split over %d GPU%s: "
                "columns %d..%d here" % (nx, nx, ncols, 16 * ncols, world, "s" if world > 1 else "", pr.first_col, pr.first_col + pr.n_cols - 1))
        return pr, "z", desc
    else:
        raise SystemExit("unknown workload " + name)
    ncols = int(pr.colIndX.max()) + 1
    if rank:  # this rank's block columns of the global system
        pr.colIndX = pr.colIndX + rank * ncols
        pr.colIndB = pr.colIndB + rank * ncols
    return pr, prec, desc


def kernel_model(pr, prec, nPairs, nA_ref, hash_in_registers=False):
    """algorithmic bytes and flops per launch of each kernel class of one iteration: (bytes the kernel has to
    move, flops, bytes by the model of SURVEY Appendix D).  The two differ for the fused multiplies of the shapes whose kernels
    recompute the shadow vector v3 from its hash in registers (16x16 z | c, 8x8 z in the default shadow mode): the model counts
    v3 (S/2 in z, S in c), the kernel never reads it -- pricing it with bytes it does not move would flatter it."""
    rb = 8 if prec == "z" else 4
    S = pr.nnzbX * 2 * pr.LM * pr.LN * rb            # one X-shaped vector
    S3 = pr.nnzbX * 2 * pr.LM * pr.LN * 4            # the float shadow vector
    A = nA_ref * 2 * pr.LM * pr.LM * rb
    idx = 4 * (pr.nnzbX + 1) + 8 * nPairs
    fm = nPairs * 8.0 * pr.LM * pr.LM * pr.LN
    el = pr.nnzbX * pr.LM * pr.LN
    S3m = 0 if hash_in_registers else S3
    return {
        "xpay_v6": (3 * S, 8.0 * el, 3 * S),
        "spmm_v4_dot": (5 * S + S3m + A + idx, fm + 24.0 * el, 5 * S + S3 + A + idx),
        "v5_nrm": (3 * S, 12.0 * el, 3 * S),
        "x_v6_v7": (7 * S, 40.0 * el, 7 * S),
        "spmm_v5_nrm_dot": (4 * S + S3m + A + idx, fm + 20.0 * el, 4 * S + S3 + A + idx),
        "multiply": (2 * S + A + idx, fm, 2 * S + A + idx),
    }


def hash_shapes(prec, LM, LN):
    """shapes whose fused multiplies recompute the hash shadow vector in registers (tfq_spmm.hip: spmm_go)"""
    return (LM, LN) == (16, 16) or (prec == "z" and (LM, LN) == (8, 8))


def S_bytes(pr, prec):
    return pr.nnzbX * 2 * pr.LM * pr.LN * (8 if prec == "z" else 4)


def roof(bytes_, flops, ms, prec):
    peak_f = MFMA_PEAK_TFLOPS["f64" if prec == "z" else "f32"]
    t_b, t_f = bytes_ / (HBM_PEAK_GBS * 1e9), flops / (peak_f * 1e12)
    gbs, tfs = bytes_ / (ms * 1e-3) / 1e9, flops / (ms * 1e-3) / 1e12
    if t_b >= t_f:
        return dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), tflops=round(tfs, 3))
    return dict(bound="mfma", achieved=round(tfs, 3), peak=peak_f, unit="TFLOP/s", frac=round(tfs / peak_f, 4), gbs=round(gbs, 1))


def cpu_baseline(pr, prec, threads):
    """The CPU path beside the GPU number, on a bounded sample of the same system: ONE tfQMR iteration + the final residual
    probe (3 of the BSR multiplies, all vector operations), timed three ways:
      * kind "reference": oracle/_ref, the reference's own CPU library (tfqmrgpu.cu compiled with HAS_NO_CUDA by
        oracle/Makefile where /root/reference existed; it travels with the tree as a built file), ONE thread -- the reference
        solver has no OpenMP (bench_tfqmrgpu.cu:358-365 is its only parallel loop);
      * value_1t: the oracle (this repository's restatement, "port") with one thread;
      * value_omp: the oracle with its multiply threaded over all cores of the CPU share.
    `value` is the reference's figure when oracle/_ref is there, the 1-thread port otherwise."""
    from oracle import pyoracle as O
    O.lib()
    an = O.analyse(pr)

    def port(nthreads):
        used = O.set_threads(nthreads)
        t0 = time.time()
        st, X, info = O.solve(pr, prec, threshold=pr.tolerance, max_iterations=1, plan=an)
        dt = time.time() - t0
        return info["flops"], dt, used
    fl, dt_omp, used = port(threads)
    _, dt_1, _ = port(1)
    O.set_threads(threads)
    out = dict(unit="TFLOP/s", value_1t=round(fl / dt_1 / 1e12, 6), seconds_1t=round(dt_1, 3),
               value_omp=round(fl / dt_omp / 1e12, 6), seconds_omp=round(dt_omp, 3), cores_omp=used, nproc=os.cpu_count())
    sample = ("1 tfQMR iteration + residual probe (3 of the BSR multiplies, all vector ops) of the same system, %.2f GFlop by the "
              "reference's count" % (fl / 1e9))
    why = "oracle/_ref not present"
    if O.have_ref():
        try:
            ref = O.Reference()
            t0 = time.time()
            st, X, info = ref.solve_staged(pr, prec, threshold=pr.tolerance, max_iterations=1)
            dt = time.time() - t0
            out.update(value=round(info["flops"] / dt / 1e12, 6), cores=1, kind="reference", seconds=round(dt, 3),
                       sample=sample + "; the reference's own CPU library (HAS_NO_CUDA build), single-threaded as the reference is")
            return out
        except AssertionError as e:
            # e.g. 65 536 block rows and more: the reference's createPlan answers 14 at tfqmrgpu.cu:169 (`nnzbA > mb*mb` in 32-bit int)
            why = "the reference's own CPU library refused this system (status %s)" % (e,)
    out.update(value=out["value_1t"], cores=1, kind="port", seconds=out["seconds_1t"],
               sample=sample + "; the oracle (restatement of the reference CPU path), one thread; " + why)
    return out


def launch_ranks(n, argv):
    """python bench.py --gpus N without a launcher: start N ranks as children of THIS process, which has not touched the GPU
    (nothing imported torch yet), relay the output, exit with the launcher's code"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + [a for a in argv if a != "--launcher"]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="fd2d_16x16_z")
    ap.add_argument("--max-iterations", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--multiply-reps", type=int, default=20)
    ap.add_argument("--no-hbm-multiply", action="store_true", help="skip the one-block-column (HBM-bound) multiply measurement")
    ap.add_argument("--no-mixed", action="store_true", help="skip the mixed-precision solve of the same system")
    ap.add_argument("--no-collectives", action="store_true", help="skip pricing the stopping test's collectives on one rank (RCCL communicator / host callback)")
    ap.add_argument("--launcher", action="store_true", help="go through torch.distributed.run even for one rank")
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or args.launcher):
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU path"
    torch.cuda.set_device(local)
    # under torch.distributed.run (RANK set) the multi-rank path is taken even for one rank, so that it can be
    # exercised on a single-GPU box: process group, RCCL communicator, all-reduced stopping test
    distributed = (world > 1) or ("RANK" in os.environ and os.environ.get("TFQMRGPU_BENCH_FORCE_DIST", "1") == "1")
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import tfqmrgpu_amd as T
    pr, prec, desc = build_problem(args.workload, rank, world)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        s = T.Solver(stream.cuda_stream)
        s.create_plan(pr)
        view = s.plan_view()
        nbytes = s.buffer_size(pr.LM, pr.LN, prec)
        buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        s.set_buffer(device_ptr=buf.data_ptr())
        s.set_matrix("A", pr.A)
        s.set_matrix("B", pr.B)
        if distributed:  # RCCL communicator of the stopping test: id from rank 0, broadcast with torch.distributed
            uid = (C.c_char * 128)()
            if rank == 0:
                T._check(T.lib.tfqmrgpuExt_commUniqueId(uid), "tfqmrgpuExt_commUniqueId")
            t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device="cuda")
            dist.broadcast(t, 0)
            uid = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().tolist()))
            ok = torch.tensor([1 if T.lib.tfqmrgpuExt_commInit(s.handle, world, rank, uid) == 0 else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            reduce_path = "rccl (library's own communicator on the solver stream, %d rank%s)" % (world, "s" if world > 1 else "")
            if int(ok.item()) == 0:
                # the library could not set up its own communicator on every rank: same protocol through the host
                # callback, reduced with torch.distributed (slower: one host round trip per stopping test)
                T.lib.tfqmrgpuExt_commDestroy(s.handle)
                reduce_path = "host callback over torch.distributed"

                def reduce_max(ctx, values, n):
                    v = torch.tensor([values[i] for i in range(n)], dtype=torch.float64, device="cuda")
                    dist.all_reduce(v, op=dist.ReduceOp.MAX)
                    v = v.cpu()
                    for i in range(n):
                        values[i] = float(v[i])
                keep_cb = T.REDUCE_CB(reduce_max)
                T._check(T.lib.tfqmrgpuExt_setReduceCallback(s.handle, keep_cb, None), "tfqmrgpuExt_setReduceCallback")

        def barrier():
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
                torch.cuda.synchronize()

        for _ in range(args.warmup):
            st = s.solve(pr.tolerance, args.max_iterations)
        def add_profile(prof):
            # [launches, ms] of the launches that did work in a steady iteration | [launches, ms] gated off | [launches, ms] of the FIRST
            # iteration of a solve (v4, v6, v7, v8, x are zero there and not read: fewer bytes, kept apart so that bytes and time match)
            first = s.profile(first=True)
            for k, (n, ms) in s.profile().items():
                a = prof.setdefault(k, [0, 0.0, 0, 0.0, 0, 0.0])
                a[0] += n - first[k][0]
                a[1] += ms - first[k][1]
                a[4] += first[k][0]
                a[5] += first[k][1]
            for k, (n, ms) in s.profile(gated=True).items():
                a = prof.setdefault(k, [0, 0.0, 0, 0.0, 0, 0.0])
                a[2] += n
                a[3] += ms

        # HIP events on the solver's stream inside the timed region bracket the two fused multiplies (the roofline kernels);
        # events around all eleven kernel classes would cost the solve 1.7 % (176 events per 16 iterations,
        # scripts/prof_overhead.py): the full breakdown comes from one further solve behind the timed region
        s.set_profiling(2)
        prof = {}
        flops = iters = 0
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st = s.solve(pr.tolerance, args.max_iterations)
            info = s.get_info()
            flops += info["flops"]
            iters += info["iterations"]
            add_profile(prof)
        barrier()
        elapsed = time.perf_counter() - t0
        s.set_profiling(1)
        prof_all = {}
        s.solve(pr.tolerance, args.max_iterations)      # (every rank: the solve holds the ranks' stopping-test all-reduces)
        add_profile(prof_all)
        barrier()
        s.set_profiling(0)
        for k in ("spmm_v4_dot", "spmm_v5_nrm_dot"):
            prof_all[k] = prof[k]
        prof = prof_all
        if distributed:
            red = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            elapsed = float(red.item())
            tot = torch.tensor([flops, float(iters)], dtype=torch.float64, device="cuda")
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            flops, iters_all = float(tot[0].item()), float(tot[1].item())
        else:
            iters_all = float(iters)

        out = None
        if rank == 0:
            nPairs = view["nPairs"]
            nA_ref = len(np.unique(view["pairs"][0::2]))
            model = kernel_model(pr, prec, nPairs, nA_ref, hash_in_registers=hash_shapes(prec, pr.LM, pr.LN))
            # avg_ms: launches that did work in steady iterations; avg_ms_all_launches also counts the launches that were enqueued ahead
            # of the stopping decision and returned at once (what a profiler's per-kernel average shows: the first-iteration launches
            # of the interleaved multiplies are kernel instances of their own, <..., FIRST = true>, and are not in either)
            for a in prof.values():      # a solve of ONE iteration has no steady launch: its first-iteration launches stand in
                if a[0] == 0 and a[4] > 0:
                    a[0], a[1] = a[4], a[5]
            per_kernel = {k: dict(launches=n, avg_ms=round(ms / n, 5), total_ms=round(ms, 3), gated_off_launches=gn,
                                  avg_ms_all_launches=round((ms + gms) / (n + gn), 5), first_iteration_launches=fn,
                                  avg_ms_first_iteration=round(fms / fn, 5) if fn else None)
                          for k, (n, ms, gn, gms, fn, fms) in prof.items() if n}
            def roof_of(k):
                r = roof(model[k][0], model[k][1], per_kernel[k]["avg_ms"], prec)
                rm_ = roof(model[k][2], model[k][1], per_kernel[k]["avg_ms"], prec)
                r.update(kernel=k, avg_ms=per_kernel[k]["avg_ms"], launches=per_kernel[k]["launches"],
                         gated_off_launches=per_kernel[k]["gated_off_launches"], avg_ms_all_launches=per_kernel[k]["avg_ms_all_launches"],
                         algorithmic_bytes=int(model[k][0]), algorithmic_flops=float(model[k][1]), traffic=None,
                         # `achieved` / `frac` price the bytes the kernel has to move; the *_model figures use SURVEY Appendix D's byte
                         # model, which counts the shadow vector even where the kernel recomputes it in registers
                         algorithmic_bytes_moved=int(model[k][0]), algorithmic_bytes_model=int(model[k][2]),
                         achieved_model=rm_["achieved"], frac_model=rm_["frac"])
                return r
            # `roofline` = the BSR multiply of the north star: of the two fused multiply kernels the one with more summed time
            # (the kernel VERDICT r01 named).  When a vector kernel has more summed time than that, it is reported next to it
            # as `roofline_dominant` -- after the multiplies went to 16-byte accesses the 7-stream update k_x_v6_v7 is level with them.
            # (every class runs once per iteration: the average per working launch ranks them, whatever number of solves it was taken over)
            dom = max((k for k in ("spmm_v4_dot", "spmm_v5_nrm_dot") if k in per_kernel), key=lambda k: per_kernel[k]["avg_ms"])
            rl = roof_of(dom)
            top = max((k for k in per_kernel if k in model), key=lambda k: per_kernel[k]["avg_ms"])
            rl_dominant = roof_of(top) if top != dom else None
            rl_all = {k: {kk: vv for kk, vv in roof_of(k).items() if kk in ("bound", "achieved", "unit", "frac", "avg_ms")} for k in per_kernel if k in model}
            # HBM bytes per launch from the PMC passes of the same command (rocprofv3 --pmc cannot run inside this timing run): a KEPT figure, quoted only
            # while it describes the kernel that runs -- every entry of profiles/pmc_traffic.json names the kernel family it was taken on
            # (scripts/pmc_to_traffic.py), the library names the family of the running plan (tfqmrgpuExt_getMultiplyKernel)
            tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            running = s.multiply_kernel()
            rl["kernel_family"] = running
            kept = json.load(open(tp)).get(args.workload) if os.path.exists(tp) else None
            if kept is None:
                rl["traffic_source"] = "null: no PMC pass of this workload is kept in profiles/pmc_traffic.json"
            elif kept.get("_kernel") != running:
                rl["traffic_source"] = "null: the kept PMC figures were taken on %s (%s, %s), this plan runs %s" % (kept.get("_kernel"), kept.get("_round"), kept.get("_sha"), running)
            elif kept.get(dom) is not None:
                rl["traffic"] = kept[dom]
                rl["traffic_over_moved"] = round(rl["traffic"] / rl["algorithmic_bytes_moved"], 3)
                rl["traffic_over_model"] = round(rl["traffic"] / rl["algorithmic_bytes_model"], 3)
                rl["traffic_source"] = ("profiles/pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE of scripts/pmc_collect.sh on %s, %s at %s (kernel %s); kept, not measured in this run"
                                        % (kept.get("_kernel"), kept.get("_round"), kept.get("_sha"), kept.get("_kernels", {}).get(dom)))

            # The BSR multiply Y = A*X on its own, twice:
            #  roofline_multiply            on the plan's data with the solver's kernel and element order (tfqmrgpuExt_applyOperator)
            #  roofline_multiply_native_api on caller-owned arrays in the reference's native order (tfqmrgpuExt_multiply: the
            #                               contract of the reference's gemmNxNf, what its `bench_tfqmrgpu multi` times)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def timed(fn, reps, batches=3):
                """ms per launch: `reps` launches back to back per batch, one warm-up batch, median of `batches` (events
                around >= 20 repetitions after the warm-ups, median)"""
                fn(reps)
                ms = []
                for _ in range(batches):
                    e0.record(stream)
                    fn(reps)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1) / reps)
                return sorted(ms)[len(ms) // 2]
            s.set_matrix("X", (np.random.default_rng(1).uniform(-1, 1, (pr.nnzbX, pr.LM, pr.LN)) + 0j)) if pr.nnzbX * pr.LM * pr.LN < 5e7 else None
            mms = timed(lambda reps=1: s.apply_operator(-reps), args.multiply_reps)    # negative: the launches alone, no copy of the product back into X
            rm = roof(model["multiply"][0], model["multiply"][1], mms, prec)
            rm.update(kernel="multiply on the plan's data (Y = A*X, no epilogue, solver's kernel and element order)", avg_ms=round(mms, 5),
                      launches=args.multiply_reps, algorithmic_bytes=int(model["multiply"][0]), algorithmic_flops=float(model["multiply"][1]))
            real = torch.float64 if prec == "z" else torch.float32
            An = torch.from_numpy(np.ascontiguousarray(np.stack([pr.A.transpose(0, 2, 1).real, pr.A.transpose(0, 2, 1).imag], axis=1))).to(real).cuda()
            Xn = torch.rand((pr.nnzbX, 2, pr.LM, pr.LN), dtype=real, device="cuda") * 2 - 1
            Yn = torch.empty_like(Xn)
            dS = torch.from_numpy(view["starts"].view(np.int32)).cuda()
            dP = torch.from_numpy(view["pairs"].view(np.int32)).cuda()

            def mult(reps=1):
                for _ in range(reps):
                    T._check(T.lib.tfqmrgpuExt_multiply(s.handle, prec.encode(), pr.LM, pr.LN, pr.nnzbX, dS.data_ptr(), dP.data_ptr(),
                                                        An.data_ptr(), Xn.data_ptr(), Yn.data_ptr()), "tfqmrgpuExt_multiply")
            mult(2)
            mms_n = timed(mult, args.multiply_reps)
            rmn = roof(model["multiply"][0], model["multiply"][1], mms_n, prec)
            rmn.update(kernel="multiply on caller-owned native arrays (tfqmrgpuExt_multiply)", avg_ms=round(mms_n, 5), launches=args.multiply_reps)
            # ... and with a launch order prepared once in front of the timed launches (tfqmrgpuExt_multiplyPrepare, mode 4: the library chooses; the
            # reference's bench prepares its launch outside its timed loop too, bench_tfqmrgpu.cu:442-556): same listing, same bits
            order = C.c_void_p(None)
            T._check(T.lib.tfqmrgpuExt_multiplyPrepare(s.handle, prec.encode(), pr.LM, pr.LN, pr.nnzbX, dS.data_ptr(), dP.data_ptr(), 4, C.byref(order)), "tfqmrgpuExt_multiplyPrepare")

            def mult_o(reps=1):
                for _ in range(reps):
                    T._check(T.lib.tfqmrgpuExt_multiplyOrdered(s.handle, prec.encode(), pr.LM, pr.LN, pr.nnzbX, dS.data_ptr(), dP.data_ptr(),
                                                               An.data_ptr(), Xn.data_ptr(), Yn.data_ptr(), order), "tfqmrgpuExt_multiplyOrdered")
            mult_o(2)
            mms_o = timed(mult_o, args.multiply_reps)
            T.lib.tfqmrgpuExt_multiplyRelease(order)
            rmn["prepared_order"] = dict(avg_ms=round(mms_o, 5), frac=roof(model["multiply"][0], model["multiply"][1], mms_o, prec)["frac"],
                                         prepared=bool(order.value), note="tfqmrgpuExt_multiplyPrepare mode 4, prepared outside the timed launches")
            del An, Xn, Yn

            # the HBM-bound corner of the multiply: the operator applied to ONE block column, every A block used once
            # (arithmetic intensity 5.8 flop/B for 16x16 z, ridge 9.8): 5-point block stencil on a 256 x 256 grid, its own plan
            rh = None
            if prec == "z" and pr.LM == 16 and pr.LN == 16 and not args.no_hbm_multiply:
                from tfqmrgpu_amd import problems as PR
                p1 = PR.stencil_2d(256, 256, 16, 16, 1, seed=11)
                s1 = T.Solver(stream.cuda_stream)
                s1.create_plan(p1)
                v1 = s1.plan_view()
                buf1 = torch.empty(s1.buffer_size(16, 16, "z"), dtype=torch.uint8, device="cuda")
                s1.set_buffer(device_ptr=buf1.data_ptr())
                s1.set_matrix("A", p1.A)
                s1.set_matrix("X", np.random.default_rng(2).uniform(-1, 1, (p1.nnzbX, 16, 16)) + 0j)
                ms1 = timed(lambda reps=1: s1.apply_operator(-reps), args.multiply_reps)
                nP1, nY1 = v1["nPairs"], p1.nnzbX
                b1 = (nP1 + 2 * nY1) * 2 * 16 * 16 * 8 + 4 * (nY1 + 1) + 8 * nP1
                f1 = nP1 * 8.0 * 16 * 16 * 16
                rh = dict(bound="hbm", achieved=round(b1 / (ms1 * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                          frac=round(b1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), tflops=round(f1 / (ms1 * 1e-3) / 1e12, 3),
                          kernel="multiply on the plan's data, operator on ONE block column (5-point block stencil 256x256, every A block used once)",
                          avg_ms=round(ms1, 5), launches=args.multiply_reps, algorithmic_bytes=int(b1), algorithmic_flops=f1)
                s1.close()
                del buf1

            # "same A, new B, solve again" (README.md:97-104 of the reference) with B and X resident on the device: setMatrix('B') + solve +
            # getMatrix('X') per right-hand side, the arrays converted in place by the library (no staging, no PCIe)
            resolve = None
            if world == 1:      # (a solve holds the ranks' collectives: never on one rank of several)
              ctype = torch.complex128 if prec == "z" else torch.complex64
              s.data_precision = prec
              dB = torch.from_numpy(pr.B).to(ctype).cuda()
              dXout = torch.empty((pr.nnzbX, pr.LM, pr.LN), dtype=ctype, device="cuda")
              torch.cuda.synchronize()
              tr0 = time.perf_counter()
              for _ in range(args.steps):
                  s.set_matrix_device("B", dB.data_ptr())
                  s.solve(pr.tolerance, args.max_iterations)
                  s.get_matrix_device(dXout.data_ptr())
              torch.cuda.synchronize()
              resolve_ms = (time.perf_counter() - tr0) / args.steps * 1e3
              # the same with pageable host arrays in the caller's own layout and precision, the C calls alone (numpy's allocation of the result and its
              # conversion to a complex array are not the library's: they had made this figure 232 ms in r03/r04 where the three calls take 55)
              hB = np.ascontiguousarray(pr.B.astype(np.complex128 if prec == "z" else np.complex64))
              Xh = np.zeros((pr.nnzbX, pr.LM, pr.LN, 2), dtype=np.float64 if prec == "z" else np.float32)
              Xh[:] = 1                                        # (pages touched: a caller's result array is not fresh from the kernel)
              pc = prec.encode()
              th0 = time.perf_counter()
              T._check(T.lib.tfqmrgpu_bsrsv_setMatrix(s.handle, s.plan, b"B", T._ptr(hB), pc, pr.LN, pr.LM, b"n", T.LAYOUT_RIRIRIRI), "setMatrix('B')")
              s.solve(pr.tolerance, args.max_iterations)
              T._check(T.lib.tfqmrgpu_bsrsv_getMatrix(s.handle, s.plan, b"X", T._ptr(Xh), pc, pr.LN, pr.LM, b"n", T.LAYOUT_RIRIRIRI), "getMatrix('X')")
              resolve_host_ms = (time.perf_counter() - th0) * 1e3
              resolve = dict(resolve_ms=round(resolve_ms, 3), over_solve=round(resolve_ms / (elapsed / args.steps * 1e3), 3),
                             resolve_host_arrays_ms=round(resolve_host_ms, 1),
                             note="setMatrix('B') + solve + getMatrix('X'); resolve_ms: B and X in device memory (converted in place by the library), "
                                  "resolve_host_arrays_ms: pageable host arrays (X = %.0f MB over PCIe), the three C calls" % (S_bytes(pr, prec) / 1e6))
              del dB, dXout, Xh, hB

            # the same system in mixed precision (bufferSize 'm': complex<float> tfQMR inside a refinement in double):
            # time to the SAME threshold in double arithmetic, beside the headline figure (which stays the complex<double> solve)
            mixed = None
            if prec == "z" and world == 1 and not args.no_mixed:
                sm = T.Solver(stream.cuda_stream)
                sm.create_plan(pr)
                mbytes = sm.buffer_size(pr.LM, pr.LN, "m")
                mbuf = torch.empty(mbytes, dtype=torch.uint8, device="cuda")
                sm.set_buffer(device_ptr=mbuf.data_ptr())
                sm.set_matrix("A", pr.A)
                sm.set_matrix("B", pr.B)
                first_its = None
                for _ in range(max(1, args.warmup)):
                    sm.solve(pr.tolerance, args.max_iterations)
                    if first_its is None:
                        first_its = sm.get_info()["iterations"]      # the plan's FIRST solve searches for the float floor of this operator; later ones remember it
                torch.cuda.synchronize()
                tm0 = time.perf_counter()
                for _ in range(args.steps):
                    stm = sm.solve(pr.tolerance, args.max_iterations)
                torch.cuda.synchronize()
                tm = (time.perf_counter() - tm0) / args.steps
                im = sm.get_info()
                sm.set_profiling(1)
                sm.solve(pr.tolerance, args.max_iterations)
                pm, pf = sm.profile(), sm.profile(first=True)
                it_m = sum((pm[k][1] - pf[k][1]) / max(1, pm[k][0] - pf[k][0]) for k in pm if k != "probe")
                mixed = dict(ms_per_solve=round(tm * 1e3, 3), solve_status=int(stm), float_iterations=im["iterations"], residual=im["residual"],
                             refinement_residuals=[float("%.3e" % v) for v in sm.refinement_history()],
                             float_iterations_per_cycle=[int(v) for v in sm.refinement_history(True)[1][:-1]],
                             ms_per_float_iteration=round(it_m, 4), buffer_GB=round(mbytes / 1e9, 3),
                             speedup_vs_double=round(elapsed / args.steps / tm, 3), float_iterations_first_solve_of_the_plan=first_its,
                             note="same threshold (max_rhs |b - A x| / |b| <= %g in double arithmetic), same system, the plan's second and later solves (same A, new B: "
                                  "the plan remembers the float floor its first solve found); not the headline metric" % pr.tolerance)
                sm.close()
                del mbuf

            # What the stopping test's collectives cost a solve BEFORE anyone runs eight GPUs: the same solves on this one rank with
            # (a) the library's RCCL communicator live (world size 1: two ncclAllReduce of 3 doubles per iteration slot on the solver's stream + the vote in
            # front of every solve) and (b) the host-callback path (an identity callback: one device -> host -> device round trip per reduction), against
            # the solves without either.  Per iteration: (ms with - ms without) / iterations.
            collectives = None
            if world == 1 and not distributed and not args.no_collectives:
                def solves_ms(n):
                    torch.cuda.synchronize()
                    tc0, its = time.perf_counter(), 0
                    for _ in range(n):
                        s.solve(pr.tolerance, args.max_iterations)
                        its += s.get_info()["iterations"]
                    torch.cuda.synchronize()
                    return (time.perf_counter() - tc0) / n * 1e3, its / n
                nsol = max(3, min(10, args.steps))
                s.solve(pr.tolerance, args.max_iterations)
                base_ms, base_it = solves_ms(nsol)
                collectives = dict(solves=nsol, ms_per_solve_no_communicator=round(base_ms, 3), iterations_per_solve=base_it)
                uid1 = (C.c_char * 128)()
                if T.lib.tfqmrgpuExt_commUniqueId(uid1) == 0 and T.lib.tfqmrgpuExt_commInit(s.handle, 1, 0, uid1) == 0:
                    s.solve(pr.tolerance, args.max_iterations)
                    ms_r, it_r = solves_ms(nsol)
                    T.lib.tfqmrgpuExt_commDestroy(s.handle)
                    collectives.update(ms_per_solve_rccl=round(ms_r, 3), iterations_per_solve_rccl=it_r,
                                       rccl_us_per_iteration=round((ms_r - base_ms) / max(1.0, it_r) * 1e3, 2))
                else:
                    collectives["rccl"] = "the library could not set up an RCCL communicator of one rank on this box"
                ident = T.REDUCE_CB(lambda ctx, values, n: None)
                if T.lib.tfqmrgpuExt_setReduceCallback(s.handle, ident, None) == 0:
                    s.solve(pr.tolerance, args.max_iterations)
                    ms_c, it_c = solves_ms(nsol)
                    T.lib.tfqmrgpuExt_setReduceCallback(s.handle, T.REDUCE_CB(0), None)
                    collectives.update(ms_per_solve_host_callback=round(ms_c, 3), iterations_per_solve_host_callback=it_c,
                                       host_callback_us_per_iteration=round((ms_c - base_ms) / max(1.0, it_c) * 1e3, 2))
                collectives["note"] = ("one rank: what the reductions of the multi-GPU protocol add to this rank's own solve (launches, the vote, no fold of small plans); "
                                       "the wire time of N > 1 ranks comes on top")

            S = pr.nnzbX * 2 * pr.LM * pr.LN * (8 if prec == "z" else 4)
            it_bytes = sum(model[k][0] for k in ("xpay_v6", "spmm_v4_dot", "v5_nrm", "x_v6_v7", "spmm_v5_nrm_dot"))
            it_ms = sum(v["avg_ms"] for k, v in per_kernel.items() if k != "probe")       # every class runs once per (steady) iteration
            out = {
                "metric": "tfQMR solve throughput to 1e-9 residual (reference flop count / solve time); iterations/s and BSR multiply GB/s + TFLOP/s in extra keys",
                "value": round(flops / elapsed / 1e12, 4), "unit": "TFLOP/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "strong" if args.workload.startswith("cfg4") else "weak", "vs_baseline": None,
                "dtype": "f64" if prec == "z" else "f32", "data": "synthetic",
                "config": {"workload": desc, "name": args.workload, "mb": pr.mb, "nnzbA": pr.nnzbA, "nnzbX_per_gpu": pr.nnzbX,
                           "block_columns_per_gpu": view["nCols"], "rhs_per_gpu": view["nCols"] * pr.LN, "pairs": nPairs,
                           "threshold": pr.tolerance, "sharding": "block columns of X/B per GPU, max-all-reduce of the stopping test",
                           "reduce_path": reduce_path if distributed else "none (one rank)"},
                "iterations_per_solve": iters / args.steps,
                "iterations_per_second": round(iters_all / world / elapsed, 2),
                "rhs_iterations_per_second": round(iters_all * view["nCols"] * pr.LN / elapsed, 1),
                "solve_status": int(st), "residual": info["residual"],
                "buffer_GB_per_gpu": round(nbytes / 1e9, 3), "vector_MB": round(S / 1e6, 1),
                "roofline": rl,
                "roofline_dominant": rl_dominant,
                "roofline_kernels": rl_all,
                "roofline_multiply": rm,
                "roofline_multiply_native_api": rmn,
                "roofline_multiply_hbm_bound": rh,
                "mixed_precision": mixed,
                "resolve": resolve,
                "collectives": collectives,
                "collective_overhead_us_per_iteration": None if collectives is None else {"rccl": collectives.get("rccl_us_per_iteration"),
                                                                                           "host_callback": collectives.get("host_callback_us_per_iteration")},
                "roofline_iteration": dict(bound="hbm", achieved=round(it_bytes / (it_ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                           frac=round(it_bytes / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), ms_per_iteration=round(it_ms, 4),
                                           algorithmic_bytes=int(it_bytes)),
                "kernels": per_kernel,
                "kernels_source": "spmm_v4_dot, spmm_v5_nrm_dot: HIP events inside the timed region (all %d solves); other classes: one further solve behind it; "
                                  "avg_ms = launches of steady iterations (the first iteration of a solve does not read the vectors that are zero there: "
                                  "avg_ms_first_iteration, fewer bytes)" % args.steps,
            }
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(pr, prec, min(16, os.cpu_count() or 1))
        s.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
