#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: MC samples/sec at batch 1e6 (camel-2D through the flow;
2->4 RAMBO) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 1000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of 1e6 synthetic samples per GPU, inputs resident in
HBM: the body of BasicManager.integrate's loop (reference manager.py:395-400) for config C2
(`PWQuadManager(n_flow=2).create_model(2, 4, [3]*3)`, camel integrand): float32 uniform w -> flow forward
with train-mode BatchNorm (what the reference's integrate() runs) -> f(X)*J -> (count, mean,
M2, max).  Samples shard across ranks (weak scaling: 1e6 per GPU); the per-rank moment records of all K steps
are combined by ONE RCCL all-reduce inside the timed region.  Rank 0 prints one JSON line.

Extra objects on the line: `roofline` (dominant kernel of the step, HIP events on the launch stream),
`cpu_baseline` (the NumPy oracle on one host core, bounded sample), `rambo` (metric part ii: 2->4 massive
RAMBO at batch 1e6 with its own roofline and CPU figure), `integral` (estimate +- sigma of the timed run).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nf_amd import _lib, integrands                                            # noqa: E402
from nf_amd.normalizing_flows.manager import PWQuadManager                     # noqa: E402
from nf_amd.parallel import allreduce_moments, init_from_env                   # noqa: E402
from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace   # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
BATCH = 1_000_000
K_CLASS = {"apply": 0, "stats": 1, "ingest": 2, "integrand": 3, "rambo": 4}


def measured_traffic(kernel, steps_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_hbm_traffic.json: FETCH_SIZE x 2 +
    WRITE_SIZE, gfx950 correction calibrated there), rescaled to this run's steps per launch; None if not profiled."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))[kernel]
        return t["bytes_per_launch"] * steps_per_launch / t["steps_per_launch"]
    except Exception:
        return None


def barrier_sync(ws):
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, ws, dev):
    if ws == 1:
        return x
    t = torch.tensor([x], dtype=torch.double, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def prof_read():
    L = _lib.lib()
    ms, n = C.c_double(0.0), C.c_int32(0)
    _lib.check(L.nis_prof_read(C.byref(ms), C.byref(n)), "nis_prof_read")
    return ms.value, n.value


def bench_flow(args, rank, ws, dev):
    torch.manual_seed(0)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, 4, [3] * 3, dev=dev.index)
    model = NF.best_model
    model.train()                                      # reference integrate() never leaves train mode
    eng = model.engine()
    f = integrands.Camel()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    G = max(1, min(args.group, args.steps))
    # resident inputs: one (G, BATCH, 2) float32 block per launch group (integrate() draws float32, manager.py:390)
    n_groups = (args.steps + G - 1) // G
    sizes = [min(G, args.steps - k * G) for k in range(n_groups)]
    draws = [torch.rand(sz, BATCH, 2, dtype=torch.float32, device=dev, generator=g) for sz in sizes]
    warm = torch.rand(G, BATCH, 2, dtype=torch.float32, device=dev, generator=g)
    out = torch.zeros((args.steps, 4), dtype=torch.double, device=dev)
    scratch = torch.zeros((G, 4), dtype=torch.double, device=dev)
    token = eng.pack_token()            # parameters do not change inside integrate(): pack the image once

    def run_group(wblock, dst):
        # ONE library call = len(wblock) steps: BN_0 records of w -> [BN stats x2 -> cell apply] x2 (cells read w's rows /
        # the planes the previous cell wrote) with camel*J + moments fused into the last apply -> one epilogue launch;
        # the steps are the y dimension of every launch.
        eng.integrate_steps(wblock, True, f.nis_kind, f.nis_params, dst, token=token)

    for _ in range(max(1, (args.warmup + G - 1) // G)):
        run_group(warm, scratch)
    if ws > 1:
        allreduce_moments(torch.zeros_like(out))        # warm-up of the job's one collective (same shape and dtype)
    L = _lib.lib()
    _lib.check(L.nis_prof_enable(K_CLASS[args.roofline_kernel], 4 * (len(draws) + 2)), "nis_prof_enable")   # <= 4 launches of a class per launch set
    barrier_sync(ws)
    t0 = time.perf_counter()
    done = 0
    for k, wblock in enumerate(draws):
        run_group(wblock, out[done:done + sizes[k]])
        done += sizes[k]
    merged = allreduce_moments(out)                     # the single collective of the job
    barrier_sync(ws)
    dt = time.perf_counter() - t0
    kern_ms, kern_n = prof_read()
    L.nis_prof_disable()
    dt = max_over_ranks(dt, ws, dev)
    mom = merged.cpu().numpy()
    var = mom[:, 2] / (mom[:, 0] - 1)
    sig = float(np.sum(mom[:, 1] / var) / np.sum(1 / var))
    sig_err = float(np.sqrt(1 / np.sum(1 / var)) / np.sqrt(mom[0, 0] * args.steps))
    sig_mc = float(1 / np.sqrt(np.sum(mom[:, 0] / var)))
    # algorithmic bytes of the dominant kernel (DESIGN.md "Kernels").  One launch covers G steps.  With float32 draws the
    # cells of the d = 2 chain read w's (N, 2) rows themselves (direct row input): cell 0 reads one 8-byte row (J_in = 1) and
    # writes y, J: 24 B/sample; the last cell reads x (plane), one row, J and -- integrand fused, X / J not stored -- writes
    # nothing: 24 B/sample.  (Plane ingest, NIS_NO_DIRECT_ROWS=1: 32 B and 24 B, mean 28 B.)
    gmean = args.steps / len(draws)
    direct = os.environ.get("NIS_NO_DIRECT_ROWS", "0") in ("", "0")
    per_launch = {"apply": gmean * BATCH * (24.0 if direct else 28.0),
                  "stats": gmean * BATCH * 8 * 1,                       # reads the P = 1 pass-through column (or its 8-byte row)
                  "ingest": gmean * BATCH * (4 * 2 if direct else 4 * 2 + 8 * 2),
                  "integrand": gmean * BATCH * 8 * (2 + 1)}[args.roofline_kernel]
    roof = None
    if kern_n > 0:
        avg_s = kern_ms / kern_n / 1e3
        ach = per_launch / avg_s / 1e9
        roof = {"kernel": {"apply": "k_cell_apply", "stats": "k_bn_stats", "ingest": "k_ingest",
                           "integrand": "k_integrand_moments"}[args.roofline_kernel],
                "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic({"apply": "k_cell_apply", "stats": "k_bn_stats", "ingest": "k_ingest"}.get(
                    args.roofline_kernel, ""), gmean),
                "avg_launch_us": round(avg_s * 1e6, 2), "launches": kern_n, "steps_per_launch": gmean,
                "algorithmic_bytes_per_launch": per_launch}
    return dict(dt=dt, samples=BATCH * args.steps * ws, sig=sig, sig_err=sig_err, sig_mc=sig_mc, roofline=roof, NF=NF)


def bench_rambo(args, rank, ws, dev):
    gen = FlatInvertiblePhasespace([100.] * 2, [100.] * 4, pdf=None, pdf_active=False)
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    r = torch.rand(BATCH, 8, dtype=torch.double, device=dev, generator=g)
    mom = torch.empty((BATCH, 6, 4), dtype=torch.double, device=dev)
    wgt = torch.empty(BATCH, dtype=torch.double, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    desc = gen._desc(1000., 0, 0, -1, False)
    for _ in range(args.warmup):
        gen._launch(desc, r, mom, wgt, status)
    L = _lib.lib()
    _lib.check(L.nis_prof_enable(K_CLASS["rambo"], args.steps + 2), "nis_prof_enable")
    barrier_sync(ws)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gen._launch(desc, r, mom, wgt, status)
    barrier_sync(ws)
    dt = time.perf_counter() - t0
    kern_ms, kern_n = prof_read()
    L.nis_prof_disable()
    dt = max_over_ranks(dt, ws, dev)
    per_launch = BATCH * (8 * 8 + 6 * 4 * 8 + 8)        # 64 B in + 192 B momenta + 8 B weight = 264 B
    avg_s = kern_ms / max(kern_n, 1) / 1e3
    ach = per_launch / avg_s / 1e9
    return {"metric": "2->4 massive RAMBO samples/sec at batch 1e6", "value": BATCH * args.steps * ws / dt,
            "unit": "samples/s", "ms_per_step": dt / args.steps * 1e3, "mean_weight": float(wgt.mean()),
            "roofline": {"kernel": "k_rambo<4>", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": measured_traffic("k_rambo<4>", 1),
                         "avg_launch_us": round(avg_s * 1e6, 2), "launches": kern_n,
                         "algorithmic_bytes_per_launch": per_launch}}


def cpu_baseline(NF):
    """The NumPy oracle (oracle/flow.py, pinned to the reference's golden vectors) on ONE host core:
    10 integrate-iterations of 1e6 samples of the same C2 workload (~8 s), plus 1e6 RAMBO points (~4 s)."""
    from oracle import flow as oflow
    from oracle import rambo as orambo
    params = {k: v.detach().cpu().numpy() for k, v in NF.best_model.state_dict().items()}
    steps = oflow.plan_pwquad(2, 2, 4)
    rng = np.random.default_rng(0)
    n_it = 10
    draws = [rng.random((BATCH, 2), dtype=np.float32) for _ in range(n_it)]
    t0 = time.perf_counter()
    sig, sig_err, *_ = oflow.integrate(oflow.camel, params, steps, draws)
    dt = time.perf_counter() - t0
    n_rb = 1_000_000
    r = rng.random((n_rb, 8))
    t1 = time.perf_counter()
    orambo.generate_kinematics([100.] * 2, [100.] * 4, 1000., r, 0, 0, -1)
    dtr = time.perf_counter() - t1
    return {"value": n_it * BATCH / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"oracle.flow.integrate: {n_it} iterations x 1e6 samples, C2 model, NumPy float64, 1 thread ({dt:.1f} s)",
            "integral": float(sig), "sigma": float(sig_err),
            "rambo": {"value": n_rb / dtr, "unit": "samples/s", "sample": f"oracle.rambo 2->4 massive, 1e6 points ({dtr:.1f} s)"},
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 200 warm-up steps (~8 ms) bring the GPU to its sustained clocks -- the apply kernel runs 128 us per launch
    # right after an idle period and 118 us in steady state -- and 1000 timed steps (~40 ms) keep start / stop costs below 1 %
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--roofline-kernel", default="apply", choices=["apply", "stats", "ingest", "integrand"])
    ap.add_argument("--group", type=int, default=10, help="steps fused into one launch set (independent BN batches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    rank, ws, local = init_from_env("nccl")
    if ws != args.gpus and ws > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ws}")
    _lib.require_gpu()
    local = local % max(1, torch.cuda.device_count())      # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    flow = bench_flow(args, rank, ws, dev)
    rambo = bench_rambo(args, rank, ws, dev)
    line = {
        "metric": "MC samples/sec at batch 1e6 (camel-2D; 2->4 RAMBO)",
        "value": flow["samples"] / flow["dt"],
        "unit": "samples/s",
        "n_gpus": ws,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": flow["dt"] / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "C2: 2-D camel, PWQuad 2 cells, n_bins 4, NN [3,3,3], train-mode BN, fused camel*J + moments",
                   "batch_per_gpu": BATCH, "global_batch": BATCH * ws, "parallelism": f"sample-shard x{ws}",
                   "collective": "one all-reduce of (steps,4) moment records"},
        "integral": {"value": flow["sig"], "sigma": flow["sig_err"], "sigma_mc": flow["sig_mc"], "truth": 0.2323235706,
                     "note": "sigma = the reference's formula (manager.py:403); sigma_mc = 1/sqrt(sum_i N/var_i)"},
        "roofline": flow["roofline"],
        "rambo": rambo,
    }
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(flow["NF"])
    if rank == 0:
        print(json.dumps(line))
    if ws > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
