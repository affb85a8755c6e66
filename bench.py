#!/usr/bin/env python3
"""bench.py — MPC-step QP solves/s of the HIP path (BASELINE.json metric).

One "step" = one pass of the hot path (lipmpc_plan_step_batch: theta/omega, closest points,
LDCBF rows, interior-point solve, certified active-set finish) over one batch of B synthetic
problems already resident in HBM.
  --gpus 1 : BASELINE configs[1]: B = 4096 robots, N = 8, 10 convex-polygon obstacles on one MI355X.
  --gpus N>1: BASELINE configs[2]: ONE batch of 32768 robots sharded contiguously over the N ranks (16384 / 8192 / 4096
              per GPU at N = 2 / 4 / 8: strong scaling of config 3), no data-path collective; RCCL only gathers the
              counters.  --total-batch / --batch override either default (--batch = per GPU, weak scaling).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP64_PEAK_DATASHEET = 78.6   # MI355X FP64 vector = matrix peak (datasheet: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz;
                             # the local microarchitecture guide lists no FP64 figure)


def fp64_peak():
    """Measured FP64 vector FMA peak of this device class (tools/fp64_peak.hip, committed result in
    profiles/r02_fp64_peak.json: 75.3 TFLOP/s with 4 waves per SIMD, 61.9 with the one wave per SIMD this kernel runs
    at); the datasheet figure only if that file is missing."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r02_fp64_peak.json")))
        return float(d["fp64_peak_tflops_measured"]), "measured (tools/fp64_peak.hip, profiles/r02_fp64_peak.json)", d
    except Exception:
        return FP64_PEAK_DATASHEET, "datasheet", None


def f_iter(n, m):
    """Algorithmic flops of one dense Mehrotra iteration."""
    return m * n * (n + 1) + n ** 3 / 3.0 + 12 * m * n + 4 * n * n + 12 * m


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def plan_launch(gpus, env):
    """What `bench.py --gpus N` has to do before anything touches the GPU (one process per GPU):
      ("run", None)    this process IS a rank (WORLD_SIZE == --gpus: the driver's torch.distributed.run form), or N = 1;
      ("spawn", cmd)   N > 1 and no WORLD_SIZE: this process is only the launcher -- it starts N fresh rank processes
                       (torch.distributed.run, rendezvous on 127.0.0.1) and relays rank 0's JSON line;
      ("error", msg)   WORLD_SIZE is set and disagrees with --gpus."""
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            return "error", f"bench.py: --gpus {gpus} but WORLD_SIZE={ws}: launch one rank per GPU (or drop WORLD_SIZE and let --gpus {gpus} start them)"
        return "run", None
    if gpus <= 1:
        return "run", None
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)]
    return "spawn", cmd


def launch_ranks(cmd, argv):
    """Parent of a `python bench.py --gpus N` run: a child process per rank (never an exec: nothing here has touched the
    GPU, and nothing will), rank 0's single JSON line relayed on stdout, the children's exit status as ours."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd + list(argv), env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    for ln in r.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return r.returncode if r.returncode != 0 else (0 if len(lines) == 1 else 3)


def make_inputs(lipmpc, synth, B, N, n_obs, lo, rank, dev, local_rank, n_fields=None, walk_steps=30, fields=None):
    """The benchmark batch: generate_obstacles-distributed fields (seed = first global index of the shard),
    delta = 0 and 0.3 variants (0.3 only where the start keeps that clearance), states = live robots of an on-device
    closed-loop warm-up of 0..walk_steps MPC steps.  n_fields < B: fields reused by several robots (different states).
    fields = (xy, nv): these fields (the committed output of the reference's own generator) instead of generated ones."""
    hi = 9.5 if N <= 8 else 15.5
    goal_xy = (10.0, 10.0) if N <= 8 else (16.0, 16.0)
    nf = n_fields or B
    xy, nv = fields if fields is not None else synth.synthetic_fields(nf, n_obs, 0.5, hi, (0.0, 0.0), goal_xy, seed=1234 + lo)
    nf = len(nv)
    if nf < B:
        rep = -(-B // nf)
        xy, nv = np.tile(xy, (rep, 1, 1, 1))[:B], np.tile(nv, (rep, 1))[:B]
    walker = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=lipmpc.FLAG_INTERIOR), local_rank)
    obs_xy = torch.as_tensor(xy, device=dev)
    obs_nv = torch.as_tensor(nv, device=dev)
    goal = torch.tensor([goal_xy], dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    st0 = torch.zeros((B, 5), dtype=torch.float64, device=dev)
    ft0 = torch.ones((B,), dtype=torch.int8, device=dev)
    ce = walker.plan_step_batch(st0, goal, ft0, obs_xy, obs_nv, None, with_c_eta=True)["c_eta"]
    clear = torch.where(obs_nv > 0, torch.linalg.norm(ce[:, :, :2], dim=2), torch.full_like(ce[:, :, 0], 1e9)).min(dim=1).values
    delta = torch.zeros((B,), dtype=torch.float64, device=dev)
    delta[B // 2:] = torch.where(clear[B // 2:] > 0.45, 0.3, 0.0)
    state, foot = synth.walk_states(walker, obs_xy, obs_nv, goal, walk_steps, seed=99 + rank, delta=delta)
    return dict(state=state, foot=foot, goal=goal, obs_xy=obs_xy, obs_nv=obs_nv, delta=delta, walker=walker)


def one_step_later(walker, state, foot, goal, obs_xy, obs_nv, delta):
    """The same robots one MPC step on (interior iterate, as a closed loop advances); a robot whose step fails stays."""
    o = walker.plan_step_batch(state, goal, foot, obs_xy, obs_nv, delta)
    st, ft = state.clone(), foot.clone()
    walker.advance(st, ft, o)
    return st.contiguous(), ft.contiguous()


def _events_ms(fn, reps, dev):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for k in range(reps):
        fn(k)
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / reps


def launch_order_figures(solver, steps, B, sa, sb, goal, obs_xy, obs_nv, delta, out, dev):
    """What the launch order is worth on this shard, outside the timed region (solves/s of this rank): index order, and the
    replayed batch (the order a launch left applied to the very same problems: an exact prediction of every cost -- the
    upper bound no closed loop gets; the timed `value` uses the one-step-stale order)."""
    both = lambda k: solver.plan_step_batch((sb if k & 1 else sa)[0], goal, (sb if k & 1 else sa)[1], obs_xy, obs_nv, delta, out=out)
    same = lambda k: solver.plan_step_batch(sa[0], goal, sa[1], obs_xy, obs_nv, delta, out=out)
    same(0); same(1)
    ms_replay = _events_ms(same, steps, dev)
    solver.set_schedule(0)
    both(0)
    ms_index = _events_ms(both, steps, dev)
    solver.set_schedule(B)
    both(0); both(1)
    return {"index_order": B / ms_index * 1e3, "replayed_batch_exact_prediction_upper_bound": B / ms_replay * 1e3}


def roofline_frac(achieved, peak):
    """`frac` is a fraction only while the dense algorithmic count stays below what the hardware can execute; the structured
    kernel executes a fraction of the dense count, so at N = 16 / 50 obstacles the ratio exceeds 1: it is then reported under
    its real name and `frac` is null (the utilisation figure is executed_frac_of_peak)."""
    r = achieved / peak
    return {"frac": r} if r <= 1.0 else {"frac": None, "dense_equiv_ratio": r}


def kept_rows(c_eta, state, delta, N, reach_step, margin=1e-3):
    """LDCBF rows that stay in each problem after the presolve (include/lipmpc.h: LIPMPC_FLAG_NO_PRESOLVE), recomputed on
    the host from the (c, eta) rows the launch reports."""
    eta, c = c_eta[:, :, 2:], c_eta[:, :, :2]
    h0 = np.einsum("bjc,bjc->bj", eta, state[:, [0, 2]][:, None, :] - c) - delta[:, None]
    pres = np.any(eta != 0.0, axis=2)
    es = np.sqrt((eta ** 2).sum(2)) * reach_step
    kept = np.zeros(h0.shape, int)
    for k in range(1, N + 1):
        kept += pres & ~(h0 > es * k + margin)
    return kept.sum(1)


def presolve_accounting(lipmpc, P, solver, inp, out, kern_ms, peak, steps, dev):
    """What the dense convention of `roofline.frac` credits once the presolve removes rows (VERDICT r3 / ADVICE): (1) the same
    batch through the kernel that keeps EVERY row (LIPMPC_FLAG_NO_PRESOLVE) -- there the dense count prices rows that are
    really in the solve: frac_no_presolve; (2) the dense count on the rows the presolve KEEPS, with this run's iterations and
    time: frac_on_kept_rows."""
    import dataclasses
    N, n_obs = P.N, P.n_obs_max
    args = (inp["state"], inp["goal"], inp["foot"], inp["obs_xy"], inp["obs_nv"], inp["delta"])
    full = lipmpc.BatchedLipMpc(dataclasses.replace(P, flags=P.flags | lipmpc.FLAG_NO_PRESOLVE), dev.index)
    o = full.alloc_outputs(args[0].shape[0])
    ms_full = _events_ms(lambda k: full.plan_step_batch(*args, out=o), max(5, steps // 2), dev)
    it_full = o["iters"].cpu().numpy().astype(np.float64)
    fl_full = f_iter(2 * N, 9 * N + N * n_obs) * float(it_full.sum())
    beta = np.sqrt(P.g / P.h_com)
    dx = max(abs(P.l_max[0]), abs(P.l_min[0])); dy = max(abs(P.l_max[1]), abs(P.l_min[1])) + abs(P.ell)
    rows = kept_rows(out["c_eta"].cpu().numpy(), inp["state"].cpu().numpy(), inp["delta"].cpu().numpy(), N, float(np.hypot(dx, dy)))
    it = out["iters"].cpu().numpy().astype(np.float64)
    fl_kept = float(np.sum(it * np.array([f_iter(2 * N, 9 * N + r) for r in rows])))
    return {"frac_no_presolve": fl_full / (ms_full * 1e-3) / 1e12 / peak, "kernel_ms_no_presolve": ms_full,
            "mean_iters_no_presolve": float(it_full.mean()), "flops_per_launch_algorithmic_no_presolve": fl_full,
            "frac_on_kept_rows": fl_kept / (kern_ms * 1e-3) / 1e12 / peak, "mean_ldcbf_rows_kept": float(rows.mean()),
            "ldcbf_rows_before_presolve": N * n_obs,
            "frac_note": "frac = SURVEY 8d dense count over ALL 9N + N n_obs rows x this run's iterations / this run's time (the bench "
                         "contract); the presolve removes most LDCBF rows before the solve, so frac credits rows the algorithm no longer "
                         "touches: frac_no_presolve = the same batch through the kernel that keeps every row (its own iterations and "
                         "time), frac_on_kept_rows = the dense count on the rows actually in the solve; executed_frac_of_peak = FP64 "
                         "instructions the hardware executed"}


def transfer_times(inp, out, dev, kern_ms):
    """Host<->device copies of ONE step's inputs and outputs through pinned host buffers (reported separately,
    never part of `value`: the timed region starts with the inputs resident in HBM)."""
    ins = [inp[k] for k in ("state", "goal", "foot", "delta", "obs_xy", "obs_nv")]
    outs = [out[k] for k in ("U", "X", "theta", "omega", "obj", "status", "iters", "active")]
    h_in = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t) for t in ins]
    h_out = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in outs]
    d_in = [torch.empty_like(t) for t in ins]
    h2d = _events_ms(lambda k: [d.copy_(h, non_blocking=True) for d, h in zip(d_in, h_in)], 10, dev)
    d2h = _events_ms(lambda k: [h.copy_(d, non_blocking=True) for h, d in zip(h_out, outs)], 10, dev)
    nb = lambda ts: int(sum(t.numel() * t.element_size() for t in ts))
    B = ins[0].shape[0]
    return {"h2d_ms": h2d, "d2h_ms": d2h, "h2d_bytes": nb(ins), "d2h_bytes": nb(outs),
            "h2d_GBps": nb(ins) / h2d / 1e6, "d2h_GBps": nb(outs) / d2h / 1e6,
            "solves_per_s_with_copies_serialised": B / (h2d + kern_ms + d2h) * 1e3,
            "note": "one step's inputs (state, goal, first_foot, delta, obstacle rings) and outputs (U, X, theta, omega, obj, status, "
                    "iters, active) through pinned host memory, copies and kernel back to back on one stream (no overlap): "
                    "what a caller that keeps nothing resident would see; not `value`"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU (weak scaling); default 4096 at --gpus 1")
    ap.add_argument("--total-batch", type=int, default=0,
                    help="problems in the whole job, sharded contiguously over the ranks (strong scaling); default 32768 at --gpus > 1")
    ap.add_argument("--horizon", type=int, default=8)
    ap.add_argument("--obstacles", type=int, default=10)
    ap.add_argument("--finish-rounds", type=int, default=0, help="lipmpc_params.finish_rounds (0 = library default)")
    ap.add_argument("--fields", type=int, default=0,
                    help="distinct obstacle fields (0 = one per robot); fewer: fields are reused by several robots (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the compact extras (BASELINE configs 3 on one GPU, 4 and 5: a few seconds; never `value`)")
    ap.add_argument("--all-configs", action="store_true",
                    help="extras in full: + config 5 through HBM rings and in closed loop (fleet of 4096 robots, 30 samples)")
    args = ap.parse_args()

    what, arg = plan_launch(args.gpus, os.environ)
    if what == "error":
        print(arg, file=sys.stderr)
        sys.exit(2)
    if what == "spawn":
        sys.exit(launch_ranks(arg, sys.argv[1:]))
    if os.environ.get("LIPMPC_BENCH_LAUNCH_ECHO") == "1":
        # launcher self-test (tests/test_sharding_gloo.py, CPU): a rank only reports how it was started
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                              "master_addr": os.environ.get("MASTER_ADDR"), "steps": args.steps}), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LIPMPC_FORCE_DIST=1: initialise RCCL even at world size 1 (rehearsal of the N>1 code path on a one-GPU box)
    use_dist = world > 1 or os.environ.get("LIPMPC_FORCE_DIST") == "1"
    # LIPMPC_BENCH_REHEARSE=1: the N>1 code path with the HIP solver on a ONE-GPU box (tests/test_sharding_gloo.py):
    # every rank on device 0, gloo instead of RCCL (two RCCL ranks cannot share a GPU).  Never a measurement.
    rehearse = os.environ.get("LIPMPC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    real_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the ONE JSON line
        # (everything else, the banner included, goes to stderr until then)
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import lipmpc
    from importlib import import_module
    synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
    sharding = import_module("humanoid-navigation-using-mpc-ldcbf_amd.sharding")

    N, n_obs = args.horizon, args.obstacles
    # workload size: config 2 on one GPU; config 3 (one batch of 32768, contiguous shards) on several
    if args.batch > 0:
        B, total, scaling, lo = args.batch, args.batch * world, "weak", rank * args.batch
    else:
        total = args.total_batch if args.total_batch > 0 else (4096 if world == 1 else 32768)
        lo, hi_ = sharding.shard_bounds(total, rank, world)
        B, scaling = hi_ - lo, ("weak" if world == 1 else "strong")
    hi = 9.5 if N <= 8 else 15.5
    goal_xy = (10.0, 10.0) if N <= 8 else (16.0, 16.0)
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, finish_rounds=args.finish_rounds)
    solver = lipmpc.BatchedLipMpc(P, local_rank)
    # more problems than the GPU holds at once (4096 at N <= 8, 2048 beyond): consecutive steps of one batch run on a cost-ordered
    # schedule (lipmpc_set_schedule: costliest first, like with like, by the previous launch's iteration counts)
    resident = 4096 if N <= 8 else 2048          # one wave per SIMD: 1024 waves of four (N <= 8) or two problems
    # ... measured with each launch placed by the costs of the SAME robots one MPC step away (what a closed loop gets, not the
    # exact prediction a replayed batch gives) the schedule is worth -2 .. +3 % (other_configs: solves_per_s_scheduled), so the
    # timed launches run in index order unless LIPMPC_BENCH_SCHEDULE=1
    scheduled = B > resident and os.environ.get("LIPMPC_BENCH_SCHEDULE") == "1"
    if scheduled:
        solver.set_schedule(B)
    walker = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=lipmpc.FLAG_INTERIOR), local_rank)

    # ---- synthetic inputs (seeded per rank), placed in HBM before the timed region -------------
    inp = make_inputs(lipmpc, synth, B, N, n_obs, lo, rank, dev, local_rank, n_fields=args.fields or None, walk_steps=30 if N <= 8 else 20)
    state, foot, goal, obs_xy, obs_nv, delta, walker = (inp[k] for k in ("state", "foot", "goal", "obs_xy", "obs_nv", "delta", "walker"))
    out = solver.alloc_outputs(B)
    # On a schedule the order a launch runs in is the one the PREVIOUS launch left.  Replaying one batch would make that an
    # exact prediction of every problem's cost, which no closed loop gets; so the timed launches alternate between the
    # batch and the same robots ONE MPC STEP LATER: each launch is placed by the costs of the neighbouring step, as the
    # samples of a closed loop are, and the workload stays stationary.
    state_b, foot_b = one_step_later(walker, state, foot, goal, obs_xy, obs_nv, delta) if scheduled else (state, foot)
    torch.cuda.synchronize(dev)

    def barrier():
        if use_dist:
            dist.barrier()

    def step(k):
        if scheduled and (k & 1):
            solver.plan_step_batch(state_b, goal, foot_b, obs_xy, obs_nv, delta, out=out)
        else:
            solver.plan_step_batch(state, goal, foot, obs_xy, obs_nv, delta, out=out)

    for k in range(args.warmup + (args.warmup & 1 if scheduled else 0)):      # (an even count: the timed loop starts on the batch)
        step(k)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    # Launch duration: ONE pair of HIP events around the K timed launches, on the stream the kernel is launched on (torch's
    # current) -- the average per launch includes the gap to the next launch, so it is an upper bound of the kernel's own
    # duration (the rocprofv3 trace in profiles/ gives that).  An event pair around EVERY launch, as rounds 1-2 had it, puts
    # two more packets between consecutive kernels and cost 10-15 us per step of the timed region itself
    # (tools/graph_gap.py: 0.116 ms per step without them against 0.131 with).
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        step(k)
    ev1.record()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    kern_ms = ev0.elapsed_time(ev1) / args.steps
    order_detail = None
    if scheduled:
        order_detail = launch_order_figures(solver, args.steps, B, (state, foot), (state_b, foot_b), goal, obs_xy, obs_nv, delta, out, dev)
        order_detail["one_step_stale_order"] = B * args.steps / elapsed
    # the answers of the batch itself, with the certificate margins the CPU check filters on (outside the timed region:
    # the identification margin costs logarithms)
    out = solver.plan_step_batch(state, goal, foot, obs_xy, obs_nv, delta, with_diag=True, with_working=True, with_c_eta=n_obs > 0)
    torch.cuda.synchronize(dev)
    status = out["status"].cpu().numpy()
    iters = out["iters"].cpu().numpy()
    n_ok = int(np.sum((status == 0) | (status == 4)))
    # RCCL all-gather of {seconds, problems, solved}: the only collective, after the timed region
    t_max, total_B, total_ok, _ = sharding.gather_counters(elapsed, B, n_ok, device=None if rehearse else dev)

    if rank == 0:
        m_rows = 9 * N + N * n_obs
        fi = f_iter(2 * N, m_rows)
        flops_launch = float(fi * iters.astype(np.float64).sum())
        achieved = flops_launch / (kern_ms * 1e-3) / 1e12
        peak, peak_src, peak_rec = fp64_peak()
        # per-launch PMC figures of this exact workload, collected by tools/profile_round.sh in separate --pmc passes
        # (HBM bytes: FETCH_SIZE / WRITE_SIZE; executed FP64 flops: SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 x 64 lanes)
        traffic = executed = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath)).get(f"N{N}_obs{n_obs}_B{B}")
                if isinstance(rec, dict):
                    traffic, executed = rec.get("hbm_bytes"), rec.get("executed_fp64_flops")
                else:
                    traffic = rec
            except Exception:
                pass
        cfg_name = ("BASELINE configs[1]" if (N, n_obs, total, world) == (8, 10, 4096, 1) else
                    "BASELINE configs[2]" if (N, n_obs, total) == (8, 10, 32768) else
                    "BASELINE configs[3]" if (N, n_obs, total) == (16, 50, 4096) else "custom")
        res = {
            "metric": f"MPC-step QP solves/sec (batch) at N={N}, {n_obs} obstacles",
            "value": total_B * args.steps / t_max,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo; not a measurement)" if rehearse else ""),
            "config": {"workload": f"{cfg_name}: {total} robots in all ({B} on rank 0), N={N}, {n_obs} convex-polygon obstacles "
                                   "(generate_obstacles distribution), states from closed-loop warm-up, delta in {0,0.3}",
                       "total_batch": total, "batch_rank0": B, "horizon": N, "obstacles": n_obs,
                       "parallelism": f"contiguous batch shards x{world}, no data-path collective",
                       "counters": dict(sharding.last_gather),
                       "launch_order": ("cost-ordered schedule (lipmpc_set_schedule), each launch placed by the costs of the same robots one "
                                        "MPC step away (the timed launches alternate between two consecutive steps of the batch)") if scheduled else "index order",
                       "launch_order_solves_per_s_rank0": order_detail},
            "solver": {"mean_iters": float(iters.mean()), "max_iters": int(iters.max()),
                       "status_hist": {str(k): int(v) for k, v in zip(*np.unique(status, return_counts=True))},
                       "solved_frac": n_ok / B},
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         **roofline_frac(achieved, peak), "traffic": traffic,
                         "kernel": "plan_step_kernel", "kernel_ms": kern_ms,
                         "kernel_ms_is": "HIP-event span of the timed region / steps (includes the gap between launches)",
                         "flops_per_launch_algorithmic": flops_launch,
                         "executed_fp64_flops_per_launch": executed,
                         "executed_tflops": (executed / (kern_ms * 1e-3) / 1e12) if executed else None,
                         "executed_frac_of_peak": (executed / (kern_ms * 1e-3) / 1e12 / peak) if executed else None,
                         "peak_source": peak_src, "peak_datasheet": FP64_PEAK_DATASHEET,
                         "frac_of_datasheet_peak_on_wall_ms_per_step": flops_launch / (t_max / args.steps) / 1e12 / FP64_PEAK_DATASHEET,
                         "sources": {"achieved": "this run: iterations of this run's answers x F_iter / kernel_ms (HIP events of this run)",
                                     "peak": peak_src,
                                     "traffic": "profiles/traffic.json (builder's rocprofv3 --pmc run of this workload, tools/profile_round.sh)",
                                     "executed_fp64_flops_per_launch": "profiles/traffic.json (same PMC run: SQ_INSTS_VALU_*_F64 x 64 lanes)"},
                         "peak_one_wave_per_simd": (peak_rec or {}).get("fp64_fma_tflops", {}).get("1_wave_per_simd"),
                         "note": "compute-bound FP64 on the vector ALU (no MFMA is issued: SQ_INSTS_VALU_MFMA_MOPS_F64 = 0); "
                                 "achieved = SURVEY 8d DENSE algorithmic count F_iter(n,m) x the iterations each problem took "
                                 "/ kernel time (HIP events on the launch stream); the kernel applies G through the "
                                 "problem's structure and EXECUTES far fewer flops: executed_* are the FP64 VALU "
                                 "instruction counters of the same launch (profiles/), a utilisation figure, whereas "
                                 "frac prices the work a dense solver would do"},
        }
        if world == 1 and n_obs > 0 and not (P.flags & (lipmpc.FLAG_INTERIOR | lipmpc.FLAG_NO_PRESOLVE)):
            res["roofline"].update(presolve_accounting(lipmpc, P, solver, inp, out, kern_ms, peak, args.steps, dev))
        if world == 1:
            res["rollout"] = rollout_throughput(lipmpc, walker, obs_xy, obs_nv, goal, delta, dev)
            rec = _traffic_record(f"rollout_N{N}_obs{n_obs}_B{B}_k40")
            if rec:      # builder's PMC passes over this very launch (tools/profile_round.sh rollout)
                ms_r = res["rollout"]["ms"]
                res["rollout"]["roofline"] = {
                    "bound": "valu_fp64", "kernel": "rollout_kernel", "kernel_ms": ms_r, "peak": peak, "unit": "TFLOP/s",
                    "achieved": f_iter(2 * N, 9 * N + N * n_obs) * res["rollout"]["mean_iters_per_step"] * res["rollout"]["mpc_steps_solved"] / (ms_r * 1e-3) / 1e12,
                    "frac": f_iter(2 * N, 9 * N + N * n_obs) * res["rollout"]["mean_iters_per_step"] * res["rollout"]["mpc_steps_solved"] / (ms_r * 1e-3) / 1e12 / peak,
                    "traffic": rec.get("hbm_bytes"), "executed_fp64_flops_per_launch": rec.get("executed_fp64_flops"),
                    "executed_frac_of_peak": (rec.get("executed_fp64_flops") or 0.0) / (ms_r * 1e-3) / 1e12 / peak,
                    "wave_alive_fraction": rec.get("wave_alive_fraction"),
                    "note": "the closed-loop kernel runs the interior mode: every present LDCBF row is in every solve, so the dense "
                            "count prices rows that are really there; traffic / executed flops: profiles/traffic.json (builder's PMC run)"}
        if world == 1 and args.all_configs:
            # the opt-in warm start of the closed loop: fewer iterations, more live state, not faster in wall time
            warm_walker = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5,
                                                                   flags=lipmpc.FLAG_INTERIOR | lipmpc.FLAG_WARM_START), local_rank)
            res["rollout_warm_start"] = rollout_throughput(lipmpc, warm_walker, obs_xy, obs_nv, goal, delta, dev)
        if world == 1:
            res["transfers"] = transfer_times(inp, out, dev, kern_ms)
        if world == 1 and not args.no_other_configs:
            res["other_configs"] = other_configs(lipmpc, synth, dev, full=args.all_configs)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(P, state, goal, foot, obs_xy, obs_nv, delta, out)
        if real_stdout is not None:
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _traffic_record(key):
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
        return rec if isinstance(rec, dict) else None
    except Exception:
        return None


def rollout_throughput(lipmpc, walker, obs_xy, obs_nv, goal, delta, dev, k_max=40):
    """Secondary figure (not `value`): the same robots walking k_max closed-loop MPC steps from rest in ONE
    launch (lipmpc_rollout_batch).  No per-step batch barrier, so a wave pays the slowest of its own 4
    robots per step instead of the slowest of the batch."""
    B = obs_xy.shape[0]
    st0 = torch.zeros((B, 5), dtype=torch.float64, device=dev)
    ft0 = torch.ones((B,), dtype=torch.int8, device=dev)
    walker.rollout(st0, goal, ft0, obs_xy, obs_nv, delta, k_max=k_max)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ro = walker.rollout(st0, goal, ft0, obs_xy, obs_nv, delta, k_max=k_max)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1)
    steps = int(ro["n_steps"].sum().item())
    return {"solves_per_s": steps / (ms * 1e-3), "ms": ms, "robots": B, "k_max": k_max, "mpc_steps_solved": steps,
            "mean_iters_per_step": float(ro["total_iters"].sum().item()) / max(steps, 1)}


def _time_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _scheduled_ms(solver, walker, inp, B, dev, reps=6):
    """ms per launch of one batch on the cost-ordered schedule, each launch placed by the costs of the same robots one MPC
    step away (see main()); and in index order."""
    sa = (inp["state"], inp["foot"])
    sb = one_step_later(walker, inp["state"], inp["foot"], inp["goal"], inp["obs_xy"], inp["obs_nv"], inp["delta"])
    o = solver.alloc_outputs(B)
    both = lambda k: solver.plan_step_batch((sb if k & 1 else sa)[0], inp["goal"], (sb if k & 1 else sa)[1], inp["obs_xy"], inp["obs_nv"],
                                            inp["delta"], out=o)
    both(0)
    ms_index = _events_ms(both, reps, dev)
    solver.set_schedule(B)
    both(0); both(1)
    ms_sched = _events_ms(both, reps, dev)
    solver.set_schedule(0)
    solver.plan_step_batch(sa[0], inp["goal"], sa[1], inp["obs_xy"], inp["obs_nv"], inp["delta"], out=o)
    torch.cuda.synchronize(dev)
    return ms_sched, ms_index, o


def other_configs(lipmpc, synth, dev, full=False):
    """Extras of the default run (never `value`): BASELINE configs[2] (32768 robots) on ONE GPU, configs[3] (N = 16, 50
    obstacles, B = 4096) and configs[4] (LiDAR front end, B = 4096), each a handful of launches.  Fields are reused by
    several robots (different states) so that the host-side generator stays short.  full: + the round-1 form of config 5
    (rings through HBM) and the closed-loop fleet."""
    out = {}
    peak = fp64_peak()[0]
    # config 2 on the REFERENCE's own obstacle stream: the 256 fields its generate_obstacles produced (committed
    # fixture tests/golden/fields_cfg2.npz, made by importing the reference), 16 robots per field at different walk steps --
    # next to the same recipe on 256 fields of the restated generator `value` uses (validated in distribution by the tests)
    gold = os.path.join(ROOT, "tests", "golden", "fields_cfg2.npz")
    if os.path.exists(gold):
        B, N, n_obs = 4096, 8, 10
        d = np.load(gold)
        rec = {}
        for name, fields in (("reference_fields", (d["rings"], d["nv"])), ("synthetic_fields_same_recipe", None)):
            inp = make_inputs(lipmpc, synth, B, N, n_obs, 0, 0, dev, dev.index, n_fields=256, fields=fields)
            solver = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5), dev.index)
            o = solver.alloc_outputs(B)
            ms = _events_ms(lambda k: solver.plan_step_batch(inp["state"], inp["goal"], inp["foot"], inp["obs_xy"], inp["obs_nv"], inp["delta"], out=o), 20, dev)
            st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
            flops = f_iter(2 * N, 9 * N + N * n_obs) * float(it.sum())
            rec[name] = {"ms_per_step": ms, "solves_per_s": B / ms * 1e3, "mean_iters": float(it.mean()), "max_iters": int(it.max()),
                         "solved_frac": float(np.isin(st, (0, 4)).mean()), "roofline_frac": flops / (ms * 1e-3) / 1e12 / peak}
        out["config2_reference_generator_fields"] = {"batch": B, "fields": 256, **rec,
                                                     "source": "tests/golden/fields_cfg2.npz = generate_obstacles of the imported reference (tests/golden/make_golden.py)"}
        del inp, solver, o
    # config 3 on one GPU: 32768 robots at N = 8 / 10 obstacles, on the schedule (8 rounds of waves)
    B, N, n_obs = 32768, 8, 10
    inp = make_inputs(lipmpc, synth, B, N, n_obs, 50000, 7, dev, dev.index, n_fields=2048)
    solver = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5), dev.index)
    ms_s, ms_i, o = _scheduled_ms(solver, inp["walker"], inp, B, dev)
    st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
    flops = f_iter(2 * N, 9 * N + N * n_obs) * float(it.sum())
    out["config3_32768_on_one_gpu"] = {"batch": B, "ms_per_step": ms_i, "solves_per_s": B / ms_i * 1e3, "launch_order": "index order",
                                       "solves_per_s_scheduled_one_step_stale": B / ms_s * 1e3,
                                       "mean_iters": float(it.mean()), "status_hist": {str(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
                                       **{("roofline_" + k): v for k, v in roofline_frac(flops / (ms_i * 1e-3) / 1e12, peak).items()}}
    del inp, solver, o
    # config 4: N = 16, 50 obstacles (streamed LDCBF rows, 32 lanes per problem), B = 4096 = two rounds of waves
    B, N, n_obs = 4096, 16, 50
    inp = make_inputs(lipmpc, synth, B, N, n_obs, 70000, 5, dev, dev.index, n_fields=512, walk_steps=20)
    solver = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5), dev.index)
    ms_s, ms_i, o = _scheduled_ms(solver, inp["walker"], inp, B, dev)
    class_counts = solver._ws[:5].cpu().numpy().tolist() if getattr(solver, "_ws", None) is not None else None
    single = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5), dev.index)
    single.auto_workspace = False        # the one dispatching kernel of rounds 1-3 (its bodies share a register allocation: scratch)
    o1 = single.alloc_outputs(B)
    ms_single = _events_ms(lambda k: single.plan_step_batch(inp["state"], inp["goal"], inp["foot"], inp["obs_xy"], inp["obs_nv"], inp["delta"], out=o1), 6, dev)
    st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
    flops = f_iter(2 * N, 9 * N + N * n_obs) * float(it.sum())
    traffic = executed = None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"N{N}_obs{n_obs}_B{B}") or {}
        traffic, executed = rec.get("hbm_bytes"), rec.get("executed_fp64_flops")
    except Exception:
        pass
    out["config4_N16_50obs"] = {"batch": B, "ms_per_step": ms_i, "solves_per_s": B / ms_i * 1e3, "launch_order": "index order",
                                "launch": "split: classify -> per-body index lists -> one kernel per solver body (lipmpc_set_workspace)",
                                "problems_per_body_1_2_4_13_25_slots": class_counts,
                                "ms_per_step_single_dispatching_kernel": ms_single, "solves_per_s_single_dispatching_kernel": B / ms_single * 1e3,
                                "solves_per_s_scheduled_one_step_stale": B / ms_s * 1e3,
                                "mean_iters": float(it.mean()), "max_iters": int(it.max()),
                                "status_hist": {str(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
                                "uncertified_frac": float((st == 4).mean()),
                                **{("roofline_" + k): v for k, v in roofline_frac(flops / (ms_i * 1e-3) / 1e12, peak).items()},
                                "executed_fp64_flops_per_launch": executed, "traffic": traffic,
                                "executed_frac_of_peak": (executed / (ms_i * 1e-3) / 1e12 / peak) if executed else None,
                                "sources": {"executed_fp64_flops_per_launch, traffic": "profiles/traffic.json (builder's rocprofv3 --pmc run, index order)"}}
    del inp, solver, o
    # config 5: LiDAR scan -> clusters -> hulls -> (c, eta) in one launch, then the step, 4096 robots on one CROWDED-style map (20 obstacles)
    B, N = 4096, 3
    exy, env = synth.synthetic_fields(1, 20, -1.0, 6.0, (-5.0, -5.0), (50.0, 50.0), seed=9, delta=0.6)
    rings = [exy[0, j, : env[0, j]] for j in range(20) if env[0, j] > 0]
    sensor = lipmpc.LidarSensor(rings, lidar_range=1.5, resolution=360, n_obs_max=12, v_max=32, device=dev.index)
    gen = torch.Generator(device=dev).manual_seed(3)
    pos = torch.rand((B, 2), dtype=torch.float64, device=dev, generator=gen) * 7.0 - 1.0
    state = torch.zeros((B, 5), dtype=torch.float64, device=dev); state[:, 0] = pos[:, 0]; state[:, 2] = pos[:, 1]
    noise = 0.01 * torch.randn((B, 360, 2), dtype=torch.float64, device=dev, generator=gen)
    goal = torch.tensor([[5.0, 5.0]], dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    foot = torch.ones((B,), dtype=torch.int8, device=dev)
    solver = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=12, v_max=32), dev.index)
    o = solver.alloc_outputs(B)
    sen = sensor.alloc_outputs(B, rings=False, c_eta=True)
    # robots handed over as they come; the call ranks them itself (estimate of the reading counts from the obstacles' bounding
    # circles -> counting sort -> launch positions, a heavy robot with light ones per SIMD: two small kernels inside the timed call, nothing carried over
    # between calls).  ms_scan_unranked: the same call without the order buffer (scans start in index order).
    sched = sensor.make_schedule(B)
    ms_scan = _time_ms(lambda: sensor.sense(state, noise, out=sen, schedule=sched))
    ms_scan_unranked = _time_ms(lambda: sensor.sense(state, noise, out=sen, schedule=None))
    ms_step = _time_ms(lambda: solver.plan_step_batch_c_eta(state, goal, foot, sen["c_eta"], None, out=o, overflow=sen["overflow"]))
    sensor.sense(state, noise, out=sen, schedule=None)
    out["config5_lidar"] = {"batch": B, "ms_scan": ms_scan, "ms_step": ms_step,
                            "robot_steps_per_s": B / (ms_scan + ms_step) * 1e3,
                            "ms_scan_unranked": ms_scan_unranked, "robot_steps_per_s_unranked": B / (ms_scan_unranked + ms_step) * 1e3,
                            "launch_order": "robots handed over in index order; the call ranks them by estimated reading count and deals them out a heavy one with light ones per SIMD (ranking included in ms_scan)",
                            "mean_inferred_obstacles": float(sen["n_inferred"].double().mean()),
                            "overflow": int(sen["overflow"].sum())}
    # config 5 on PER-ROBOT maps of the reference's unknown-environment scenario shape (Scenario.load_scenario(CROWDED, start (0,0),
    # goal (4,3.5), 20 obstacles, range (-1,6)^2, delta 1: simulation_1.py:195-232): 256 distinct maps, 16 robots each, every
    # robot somewhere in its map's box outside the obstacles (clearance 0.05)
    n_maps, n_env = 256, 20
    mxy, mnv = synth.synthetic_fields(n_maps, n_env, -1.0, 6.0, (0.0, 0.0), (4.0, 3.5), seed=77, delta=1.0)
    rng = np.random.default_rng(5)
    pos_h = np.zeros((B, 2))
    for b in range(B):
        m = b % n_maps
        polys = [mxy[m, j, : mnv[m, j]] for j in range(n_env) if mnv[m, j] > 0]
        while True:
            p = rng.uniform(-1.0, 6.0, 2)
            if all(synth._dist_point_poly(p, q) > 0.05 and not synth._inside(p, q) for q in polys):
                break
        pos_h[b] = p
    env_xy = torch.as_tensor(np.tile(mxy, (B // n_maps, 1, 1, 1)), device=dev).contiguous()
    env_nv = torch.as_tensor(np.tile(mnv, (B // n_maps, 1)), device=dev).contiguous()
    state_m = torch.zeros((B, 5), dtype=torch.float64, device=dev)
    state_m[:, 0] = torch.as_tensor(pos_h[:, 0], device=dev); state_m[:, 2] = torch.as_tensor(pos_h[:, 1], device=dev)
    goal_m = torch.tensor([[4.0, 3.5]], dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    ms_scan_m = _time_ms(lambda: sensor.sense(state_m, noise, out=sen, schedule=sched, env_xy=env_xy, env_nv=env_nv))
    ms_scan_m_unranked = _time_ms(lambda: sensor.sense(state_m, noise, out=sen, schedule=None, env_xy=env_xy, env_nv=env_nv))
    ms_step_m = _time_ms(lambda: solver.plan_step_batch_c_eta(state_m, goal_m, foot, sen["c_eta"], None, out=o, overflow=sen["overflow"]))
    stm = o["status"].cpu().numpy()
    out["config5_lidar_per_robot_maps"] = {"batch": B, "maps": n_maps, "obstacles_per_map_mean": float((mnv > 0).sum(1).mean()),
                                           "ms_scan": ms_scan_m, "ms_scan_unranked": ms_scan_m_unranked, "ms_step": ms_step_m,
                                           "robot_steps_per_s": B / (ms_scan_m + ms_step_m) * 1e3,
                                           "mean_inferred_obstacles": float(sen["n_inferred"].double().mean()), "overflow": int(sen["overflow"].sum()),
                                           "status_hist": {str(k): int(v) for k, v in zip(*np.unique(stm, return_counts=True))},
                                           "maps_like": "Scenario.load_scenario(CROWDED, (0,0), (4,3.5), 20, range (-1,6)^2): simulation_1.py:195-232",
                                           "launch_order": "as config5_lidar"}
    sensor.sense(state, noise, out=sen, schedule=None)
    if not full:
        return out
    # the two-launch form of round 1 for comparison: rings through HBM, geometry front end in the step kernel
    sen_r = sensor.alloc_outputs(B)
    ms_scan_r = _time_ms(lambda: sensor.sense(state, noise, out=sen_r, schedule=None))
    ms_step_r = _time_ms(lambda: solver.plan_step_batch(state, goal, foot, sen_r["obs_xy"], sen_r["obs_nv"], None, out=o))
    out["config5_lidar"]["rings_through_hbm"] = {"ms_scan": ms_scan_r, "ms_step": ms_step_r}
    # config 5 in closed loop: the same fleet walking 30 samples through the map, one captured HIP graph per sample
    fleet = lipmpc.UnknownEnvFleet(rings, N_horizon=N, lidar_range=1.5, resolution=360, n_obs_max=12, v_max=32, device=dev.index)
    st0 = torch.zeros((B, 5), dtype=torch.float64, device=dev)
    st0[:, 0] = -1.8 + 0.5 * torch.rand((B,), dtype=torch.float64, device=dev, generator=gen)
    st0[:, 2] = -1.5 + 7.5 * torch.rand((B,), dtype=torch.float64, device=dev, generator=gen)
    K = 30
    fleet.run(st0, goal, foot, K, noise_seed=4)          # first run of this shape: buffers + graph capture
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    r = fleet.run(st0, goal, foot, K, noise_seed=5)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    solved = int(r["n_steps"].sum())
    out["config5_closed_loop"] = {"robots": B, "samples": K, "ms": dt * 1e3, "robot_steps_solved": solved,
                                  "robot_steps_per_s": solved / dt, "robots_walking_at_end": int((r["n_steps"] == K).sum()),
                                  "overflow_samples": int(r["overflow"].sum())}
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(P, state, goal, foot, obs_xy, obs_nv, delta, out):
    """The C oracle (dense port of the same algorithm) timed on this box's host cores on a bounded
    sample of the same workload.  Reported baseline only; it also re-checks the GPU result."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    h = lambda t: t.cpu().numpy()
    st, go, fo, xy, nv, de = h(state), h(goal), h(foot), h(obs_xy), h(obs_nv), h(delta)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # a container's CPU quota (cgroup v2) is the real core budget
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    cores = min(cores, 64)
    B = st.shape[0]
    t0 = time.perf_counter()
    r1 = c_oracle.plan_step_batch(P, st, go, fo, xy, nv, de, n_threads=1)
    t1 = time.perf_counter() - t0
    reps, t_all, done = 0, 0.0, 0
    while t_all < 8.0 and reps < 200:
        t0 = time.perf_counter()
        c_oracle.plan_step_batch(P, st, go, fo, xy, nv, de, n_threads=cores)
        t_all += time.perf_counter() - t0
        reps += 1
        done += B
    U = out["U"].cpu().numpy()
    ok = (r1["status"] == 0) & (out["status"].cpu().numpy() == 0)
    du = float(np.max(np.abs(U[ok] - r1["U"][ok]))) if ok.any() else float("nan")
    # UNCERTIFIED answers (interior-point iterate handed out as usable) against the optimum the oracle certifies when its
    # finish may run 64 rounds
    unc = np.where(out["status"].cpu().numpy() == 4)[0]
    du_unc, n_unc_cert = None, 0
    if len(unc):
        import dataclasses
        P64 = dataclasses.replace(P, finish_rounds=64)
        r64 = c_oracle.plan_step_batch(P64, st[unc], go[unc], fo[unc], xy[unc], nv[unc], de[unc], n_threads=cores)
        c64 = r64["status"] == 0
        n_unc_cert = int(c64.sum())
        du_unc = float(np.max(np.abs(U[unc][c64] - r64["U"][c64]))) if c64.any() else None
    # active sets, bit for bit, by the parity tests' own check (tests/helpers.py::compare_active_sets): `active` is the primal
    # tight set of the optimum (slack <= 1e-7, unique), compared on every certified problem but those with a row within 10 x the
    # distance between the two answers of that tolerance; the finish's working sets where both certificates are decisive
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import compare_active_sets
    act_info, _ = compare_active_sets(ok, {k: out[k].cpu().numpy() for k in ("U", "X", "active", "working", "diag")}, r1)
    return {"value": done / t_all, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"the same {B}-problem batch x {reps} passes, OpenMP over problems ({cores} threads); "
                      f"single thread: {B / t1:.0f} solves/s",
            "value_1thread": B / t1, "cpu_model": _cpu_model(),
            "max_abs_dU_gpu_vs_cpu": du,
            "status_mismatches": int(np.sum(r1["status"] != out["status"].cpu().numpy())),
            "active_set_mismatches": act_info["active_mismatch"], "active_sets_compared": act_info["active_compared"],
            "active_sets_certified_both": int(ok.sum()), "active_sets": act_info,
            "uncertified": int(len(unc)), "uncertified_certified_by_64_round_oracle": n_unc_cert,
            "max_abs_dU_uncertified_vs_certified_optimum": du_unc}


if __name__ == "__main__":
    main()
