#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: MC samples/sec at batch 1e6 (camel-2D through the flow;
2->4 RAMBO) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 1000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of 1e6 synthetic samples per GPU: the WHOLE body of
BasicManager.integrate's loop (reference manager.py:395-400) for config C2
(`PWQuadManager(n_flow=2).create_model(2, 4, [3]*3)`, camel integrand): draw the float32 uniform batch (the draw is part
of the loop, :395 -- here the first launch of a step generates it on the device, Philox4x32-10), flow forward with
train-mode BatchNorm (what the reference's integrate() runs), f(X)*J, (count, mean, M2, max).  Nothing is read
from HBM that the step did not write itself.  Steps are issued as launch sets of `--group` steps (independent BatchNorm
batches in the y dimension of every launch; a set that finds its stream idle is replayed from a hipGraph).  Samples shard
across ranks (weak scaling: 1e6 per GPU); the per-rank moment records of all K steps are combined by ONE RCCL all-reduce
inside the timed region.  Rank 0 prints one JSON line.

Extra objects on the line: `roofline` (dominant kernel of the step: fp64 VALU issue, which is what binds it; HIP events on
the launch stream + the committed PMC instruction counts), `roofline_hbm` (the same kernel and the whole step against HBM),
`cpu_baseline` (the PyTorch-CPU restatements of the flow and of the 2->4 map on all host cores + the NumPy oracles on one), `rambo` (metric part ii),
`configs` (BASELINE configs C3 / C4 / C5 at their per-GPU sizes, matrix-core path), `integral` (estimate +- sigma).
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
from nf_amd.normalizing_flows.manager import PWLinManager, PWQuadManager       # noqa: E402
from nf_amd.parallel import allreduce_moments, init_from_env                   # noqa: E402
from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace   # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
VALU_F64_PEAK = 256 * 4 * 16 * 2.4e9     # fp64 vector issue: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 3.93e13 lane-instr/s (78.6 TFLOP/s FMA)
MFMA_F64_PEAK_TF = 78.6        # v_mfma_f64_16x16x4_f64: same datapath as the fp64 vector pipe
BATCH = 1_000_000
ALG_BYTES_C2 = 16.0            # fused integrate reads w only, 8 d bytes per sample
ALG_FLOP_CELL_C2 = 180.0       # 2 (3 + 9 + 9 + 27) + ~80 flop per cell and sample
K_CLASS = {"apply": 0, "stats": 1, "ingest": 2, "integrand": 3, "rambo": 4}


def profile_json(name, with_name=False):
    """The newest committed rocprofv3 PMC summary (profiles/rNN_<name>.json, written by tools/profile_round*.sh), or {}."""
    for tag in ("r04", "r03", "r02", "r01"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_{name}.json")))
            return (d, f"{tag}_{name}.json") if with_name else d
        except Exception:
            continue
    return ({}, "") if with_name else {}


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


def combine(mom, n_per_batch):
    """manager.py:402-403 on (steps, 4) records; also the plain Monte-Carlo error of the combined estimate."""
    var = mom[:, 2] / (mom[:, 0] - 1)
    sig = float(np.sum(mom[:, 1] / var) / np.sum(1 / var))
    sig_err = float(np.sqrt(1 / np.sum(1 / var)) / np.sqrt(n_per_batch * mom.shape[0]))
    sig_mc = float(1 / np.sqrt(np.sum(mom[:, 0] / var)))
    return sig, sig_err, sig_mc


def bench_flow(args, rank, ws, dev):
    torch.manual_seed(0)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, 4, [3] * 3, dev=dev.index)
    model = NF.best_model
    model.train()                                      # reference integrate() never leaves train mode
    eng = model.engine()
    f = integrands.Camel()
    G = max(1, min(args.group, args.steps))
    plan = eng.plan(G, BATCH, dev, f.nis_kind, f.nis_params)      # one captured launch set (the capture runs a 512-sample set)
    seed = 1234 + rank
    plan.reset(seed, 0)
    # the launch set is captured AFTER the process group is up (init_from_env ran first): with NIS_BENCH_REQUIRE_GRAPH=1 (multi-rank
    # rehearsals) an eager fallback of the capture is an error instead of a silently different timed region
    if os.environ.get("NIS_BENCH_REQUIRE_GRAPH") and not plan.captured:
        raise SystemExit("bench.py: the launch set was not captured into a hipGraph (eager fallback) although NIS_BENCH_REQUIRE_GRAPH is set")
    out = torch.zeros((args.steps, 4), dtype=torch.double, device=dev)
    scratch = torch.zeros((max(G, args.warmup, 1), 4), dtype=torch.double, device=dev)
    tok = plan.token

    def run_steps(n, dst, stream0):
        # ONE library call = n steps: [draw + BN_0 records] -> [BN stats x2 -> cell apply] x2 with camel*J + moments fused into
        # the last apply -> one epilogue launch; the steps are the y dimension of every launch.
        eng.integrate_steps_rng(n, BATCH, dev, f.nis_kind, f.nis_params, dst, seed, stream0, token=tok)

    if args.warmup > 0:
        run_steps(args.warmup, scratch[:args.warmup], 1 << 40)     # exactly W untimed steps (their own generator streams)
    if ws > 1:
        allreduce_moments(torch.zeros_like(out))        # warm-up of the job's one collective (same shape and dtype)
    barrier_sync(ws)
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:
        n = min(G, args.steps - done)
        if n == G:
            plan.run(out[done:done + n])                # graph replay when the stream is idle, eager launches otherwise
        else:
            run_steps(n, out[done:done + n], done)
        done += n
    merged = allreduce_moments(out)                     # the single collective of the job
    barrier_sync(ws)
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, ws, dev)
    sig, sig_err, sig_mc = combine(merged.cpu().numpy(), BATCH * ws)

    # ---- roofline of the dominant kernel: one more EAGER set with HIP events around every launch of the class (a graph
    #      replay cannot carry them); instruction counts / bytes per launch from the committed PMC passes of the same set size
    L = _lib.lib()
    _lib.check(L.nis_prof_enable(K_CLASS[args.roofline_kernel], 16), "nis_prof_enable")
    run_steps(G, scratch[:G], 1 << 41)
    torch.cuda.synchronize()
    kern_ms, kern_n = prof_read()
    L.nis_prof_disable()
    pmc, pmc_file = profile_json("pmc_c2", with_name=True)   # {"steps_per_launch": g, "kernels": {name: {"valu_insts":, "hbm_bytes":, "launches":}}}
    roof = roof_hbm = roof_issue = None
    if kern_n > 0:
        avg_s = kern_ms / kern_n / 1e3
        kname = {"apply": "k_cell_apply", "stats": "k_bn_stats_fold", "ingest": "k_ingest2_rng"}.get(args.roofline_kernel, "")
        rec = pmc.get("kernels", {}).get(kname, {})
        scale = G / max(1, pmc.get("steps_per_launch", G))
        units = BATCH * G                                   # samples one launch of the kernel processes (one cell of G steps)
        # ALGORITHMIC roofline: the fused step reads 8 d = 16 B/sample (nothing at all with the draw inside) and
        # does 180 flop per cell and sample; one apply launch = one cell of `units` samples.  The larger fraction names the bound.
        ach_hbm = ALG_BYTES_C2 * units / avg_s / 1e9
        ach_tf = ALG_FLOP_CELL_C2 * units / avg_s / 1e12
        f_hbm, f_flop = ach_hbm / HBM_PEAK_GBS, ach_tf / MFMA_F64_PEAK_TF
        traffic = rec.get("hbm_bytes") and rec["hbm_bytes"] * scale
        common = {"kernel": kname, "avg_launch_us": round(avg_s * 1e6, 2), "launches": kern_n, "steps_per_launch": G, "units_per_launch": units}
        roof_hbm = dict(common, bound="hbm", achieved=round(ach_hbm, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(f_hbm, 4), traffic=traffic,
                        algorithmic_bytes_per_sample=ALG_BYTES_C2)
        roof_flop = dict(common, bound="valu_f64", achieved=round(ach_tf, 2), peak=MFMA_F64_PEAK_TF, unit="TFLOP/s", frac=round(f_flop, 4),
                         traffic=traffic, algorithmic_flop_per_sample_and_cell=ALG_FLOP_CELL_C2,
                         note="narrow (width-3) conditioners run on the fp64 vector pipe, whose peak equals the dense f64 MFMA peak (78.6 TFLOP/s)")
        roof = roof_flop if f_flop >= f_hbm else roof_hbm
        roof["other_bound"] = {"bound": "hbm" if roof is roof_flop else "valu_f64", "frac": round(f_hbm if roof is roof_flop else f_flop, 4)}
        # whole step against HBM, with the PMC traffic of all kernels of a set
        step_bytes = sum(k.get("hbm_bytes", 0) * k.get("launches_per_set", 1) for k in pmc.get("kernels", {}).values())
        per_sample = step_bytes / (pmc.get("steps_per_launch", G) * BATCH) if step_bytes else None
        roof_hbm.update(step_traffic_B_per_sample=per_sample and round(per_sample, 1), traffic_ratio=per_sample and round(per_sample / ALG_BYTES_C2, 2),
                        step_frac=round(ALG_BYTES_C2 * BATCH * args.steps / dt / 1e9 / HBM_PEAK_GBS, 4),
                        step_flop_frac=round(2 * ALG_FLOP_CELL_C2 * BATCH * args.steps / dt / 1e12 / MFMA_F64_PEAK_TF, 4))
        # DIAGNOSIS (not the roofline): fp64 issue-slot utilisation = lane-instructions per launch (SQ_INSTS_VALU x 64, counted in the
        # committed PMC pass named in insts_from, not in this run) / the duration measured here
        if rec.get("valu_insts"):
            lane = rec["valu_insts"] * 64.0 * scale
            roof_issue = dict(common, bound="valu_f64_issue", achieved=round(lane / avg_s / 1e9, 1), peak=round(VALU_F64_PEAK / 1e9, 1),
                              unit="Glane-instr/s", frac=round(lane / avg_s / VALU_F64_PEAK, 4), insts_from="profiles/" + pmc_file,
                              valu_insts_per_sample=round(rec["valu_insts"] * 64.0 / (pmc.get("steps_per_launch", G) * BATCH), 1),
                              note="both apply launches of a set (cell 0, last cell with the fused camel) averaged")
    return dict(dt=dt, samples=BATCH * args.steps * ws, sig=sig, sig_err=sig_err, sig_mc=sig_mc, roofline=roof, roofline_hbm=roof_hbm,
                roofline_issue=roof_issue, G=G, graph=bool(plan.captured))


def bench_rambo(args, rank, ws, dev):
    gen = FlatInvertiblePhasespace([100.] * 2, [100.] * 4, pdf=None, pdf_active=False)
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    NSET = 4                                               # buffer sets in rotation: 4 x 264 MB > the 256 MiB Infinity Cache
    r = [torch.rand(BATCH, 8, dtype=torch.double, device=dev, generator=g) for _ in range(NSET)]
    mom = [torch.empty((BATCH, 6, 4), dtype=torch.double, device=dev) for _ in range(NSET)]
    wgt = [torch.empty(BATCH, dtype=torch.double, device=dev) for _ in range(NSET)]
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    desc = gen._desc(1000., 0, 0, -1, False)
    for k in range(args.warmup):
        gen._launch(desc, r[k % NSET], mom[k % NSET], wgt[k % NSET], status)
    L = _lib.lib()
    _lib.check(L.nis_prof_enable(K_CLASS["rambo"], args.steps + 2), "nis_prof_enable")
    barrier_sync(ws)
    t0 = time.perf_counter()
    for k in range(args.steps):
        gen._launch(desc, r[k % NSET], mom[k % NSET], wgt[k % NSET], status)
    barrier_sync(ws)
    dt = time.perf_counter() - t0
    kern_ms, kern_n = prof_read()
    L.nis_prof_disable()
    dt = max_over_ranks(dt, ws, dev)
    per_launch = BATCH * (8 * 8 + 6 * 4 * 8 + 8)        # 64 B in + 192 B momenta + 8 B weight = 264 B
    avg_s = kern_ms / max(kern_n, 1) / 1e3
    ach = per_launch / avg_s / 1e9
    wall = torch.cat(wgt[:min(NSET, max(1, args.steps))])
    vol, verr = float(wall.mean()), float(wall.std() / np.sqrt(wall.numel()))
    pmc = profile_json("pmc_rambo").get("kernels", {}).get("k_rambo", {})
    return {"metric": "2->4 massive RAMBO samples/sec at batch 1e6", "value": BATCH * args.steps * ws / dt,
            "unit": "samples/s", "ms_per_step": dt / args.steps * 1e3,
            "integral": {"value": vol, "sigma": verr, "what": "phase-space volume <w> of 2->4 at E_cm = 1000, m = 100 (unit matrix element); "
                         "reference probe 0.02482 (SURVEY 4)", "samples": int(wall.numel())},
            "roofline": {"kernel": "k_rambo<4>", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc.get("hbm_bytes"),
                         "avg_launch_us": round(avg_s * 1e6, 2), "launches": kern_n, "buffer_sets": NSET,
                         "algorithmic_bytes_per_launch": per_launch}}


CONFIGS = {
    # name: (manager, d, create_model kwargs, per-GPU batch, GPUs the config is quoted on, flop/sample of SURVEY 8(d))
    "C3": ("pwlin", 8, dict(n_pass_through=4, n_cells=4, n_bins=64, NN=[32] * 3, roll_step=4), 4_000_000, 1, 8.3e4),
    "C4": ("pwquad", 10, dict(n_cells=5, n_bins=16, NN=[32] * 3), 500_000, 4, 1.2e5),
    "C5": ("pwquad", 16, dict(n_cells=8, n_bins=16, NN=[32] * 3), 1_000_000, 8, 1.9e5),
}


def config_integrand(name):
    if name == "C3":
        return integrands.Normal(0.5, 0.1, 8), "normal(x, 0.5, 0.1, 8)"
    if name == "C4":
        gen = FlatInvertiblePhasespace([0., 0.], [100.] * 4, pdf=None, pdf_active=True)
        return integrands.RamboWeight(gen, 13000., 0, 0, -1, [2, -1]), "RAMBO 2->4 massive weight, tau/y sampling, E_cm 13000"
    gen = FlatInvertiblePhasespace([0., 0.], [0.] * 6, pdf=None, pdf_active=True)
    return integrands.RamboWeight(gen, 13000., 20, 0.4, 2.5, [21, 21]), "RAMBO 2->6 massless weight, cuts pT 20 / dR 0.4 / eta 2.5"


def bench_config(name, rank, ws, dev, steps=6):
    """BASELINE config `name` at its per-GPU batch: `steps` iterations of integrate()'s loop body (draw, flow forward with
    train-mode BN on the f64 matrix cores, integrand * J, moments), records of all ranks combined by one all-reduce."""
    kind, d, kw, N, quoted_gpus, flops = CONFIGS[name]
    torch.manual_seed(10 + rank)
    NF = (PWQuadManager if kind == "pwquad" else PWLinManager)(n_flow=d)
    NF.create_model(**kw) if kind == "pwlin" else NF.create_model(dev=dev.index, **kw)
    NF.best_model.train()
    f, fdesc = config_integrand(name)
    NF.moments_batches(f, steps, N, dev)                # warm-up: the same call (code objects, workspaces and scratch planes sized for it)
    barrier_sync(ws)
    t0 = time.perf_counter()
    mom = NF.moments_batches(f, steps, N, dev)
    merged = allreduce_moments(mom)
    barrier_sync(ws)
    dt = max_over_ranks(time.perf_counter() - t0, ws, dev)
    sig, sig_err, _ = combine(merged.cpu().numpy(), N * ws)
    rate = N * steps * ws / dt
    # ---- dominant kernel alone: the same call once more with HIP events around every launch of the apply class (the launches of a
    #      wide config are eager, not a graph replay).  All of SURVEY 8(d)'s algorithmic flop sits in the apply launches (one per cell and
    #      launch set; the statistics launches recompute), so kernel-only achieved = flop/sample x samples / the apply launches' time.
    L = _lib.lib()
    _lib.check(L.nis_prof_enable(K_CLASS["apply"], 4096), "nis_prof_enable")
    NF.moments_batches(f, steps, N, dev)
    torch.cuda.synchronize()
    kern_ms, kern_n = prof_read()
    L.nis_prof_disable()
    pmc, pmc_file = profile_json("pmc_wide", with_name=True)
    pmc = pmc.get(name, {})
    tf_step = flops * rate / ws / 1e12
    roof = {"kernel": "k_cell_apply_mfma", "bound": "mfma_f64", "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s", "flop_per_sample": flops,
            "mfma_util": pmc.get("mfma_util"), "traffic": pmc.get("hbm_bytes_per_step"),
            "step_achieved": round(tf_step, 2), "step_frac": round(tf_step / MFMA_F64_PEAK_TF, 4)}
    if kern_n > 0 and kern_ms > 0:
        apply_ms_per_step = kern_ms / steps
        tf_k = flops * N / (apply_ms_per_step * 1e-3) / 1e12
        roof.update(achieved=round(tf_k, 2), frac=round(tf_k / MFMA_F64_PEAK_TF, 4), apply_ms_per_step=round(apply_ms_per_step, 4),
                    apply_launches_per_step=round(kern_n / steps, 2), apply_share_of_step=round(apply_ms_per_step / (dt / steps * 1e3), 3),
                    note="achieved / frac = SURVEY 8(d) algorithmic flop per sample x samples / the apply launches' HIP-event time (kernel only); "
                         "step_* = the same flop over the WHOLE step's wall time")
        # ATTAINABLE for this instruction stream: on gfx950 a f64 MFMA (64 cycles) and every VALU instruction (4 cycles per wave) take SIMD
        # time one after the other whichever wave issues them (profiles/r04_mfma_valu_overlap.txt), so a launch cannot finish before
        # (64 #MFMA + 4 #VALU) / (1024 SIMDs x 2.4 GHz).  Counts per launch from the committed PMC pass (not counted in this run).
        ak = pmc.get("kernels", {}).get("k_cell_apply_mfma", {})
        if ak.get("mfma_insts") and ak.get("valu_insts") and ak.get("samples_per_launch"):
            n_cells = len(NF.best_model.engine().cells)
            per = N * steps * n_cells / max(kern_n, 1)                         # samples ONE apply launch of this run covers: a launch set's iterations x N (one launch per cell and set)
            scale = per / ak["samples_per_launch"]
            cyc = (64.0 * ak["mfma_insts"] + 4.0 * (ak["valu_insts"] - ak["mfma_insts"])) * scale
            t_att = cyc / (1024 * 2.4e9)
            t_meas = kern_ms / kern_n * 1e-3
            roof["attainable"] = {"t_mfma_plus_valu_us": round(t_att * 1e6, 1), "measured_us": round(t_meas * 1e6, 1),
                                  "frac_of_attainable": round(t_att / t_meas, 4), "mfma_share": round(64.0 * ak["mfma_insts"] * scale / cyc, 3),
                                  "insts_from": "profiles/" + pmc_file,
                                  "note": "t(MFMA) + t(VALU) of the apply kernel's own instruction stream / its measured launch time"}
    else:
        roof.update(achieved=round(tf_step, 2), frac=round(tf_step / MFMA_F64_PEAK_TF, 4), note="whole step (no per-kernel events)")
    return {"workload": f"{name}: {kind} d={d} {kw}, {fdesc}", "batch_per_gpu": N, "quoted_on_gpus": quoted_gpus,
            "value": rate, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "integral": {"value": sig, "sigma": sig_err}, "roofline": roof}


TRAIN_CASES = {
    # "integrate(...) and one train step with mbs = 1e6"; C5 topology at the minibatch the adjoint's workspace is sized for;
    # C1_epoch: one optimizer step of the reference's DEFAULT regime (manager.py:66-69: batch_size 10000, mini_batch_size 2000 -> five
    # minibatches per backward), which is launch-bound: the five minibatches travel as grid.y of every launch
    "C2": ("C2", 1_000_000, 1, 360.0),        # (model, minibatch size, minibatches per optimizer step, forward flop per sample: 2 cells x 180)
    "C5": ("C5", 200_000, 1, 1.9e5),
    "C1_epoch": ("C2", 2_000, 5, 360.0),
    # the C2 step is host-bound at 1e6 (~0.9 ms of Python / autograd / optimizer against ~0.5 ms of kernels); at 4e6 the kernels decide
    "C2_mbs4e6": ("C2", 4_000_000, 1, 360.0),
    # opt-in captured step (NIS_TRAIN_GRAPH=1, flow.TrainPlan: the step's library calls as two hipGraph replays; capture ~0.2 s, untimed)
    "C2_graph": ("C2", 1_000_000, 1, 360.0),
    "C1_epoch_graph": ("C2", 2_000, 5, 360.0),
}


def bench_train_step(name, rank, ws, dev, steps=20):
    """One training minibatch (reference manager.py:219-280 with n_minibatches = 1): draw w, train-mode forward with history, integrand,
    loss bookkeeping (nis_loss_terms), hand-written adjoint seeded by the loss, optimizer step.  `steps` timed after 5 warm-up steps."""
    from nf_amd.normalizing_flows import _autograd
    from nf_amd.normalizing_flows._autograd import flow_loss_batched
    cfg, mbs, G, fwd_flop = TRAIN_CASES[name]
    _autograd.TRAIN_GRAPH = name.endswith("_graph")
    if G > 1:
        steps = 50
    torch.manual_seed(20 + rank)
    if cfg == "C2":
        NF = PWQuadManager(n_flow=2)
        NF.create_model(2, 4, [3] * 3, dev=dev.index)
        f, d = integrands.Camel(), 2
    else:
        kind, d, kw, _, _, _ = CONFIGS[cfg]
        NF = PWQuadManager(n_flow=d)
        NF.create_model(dev=dev.index, **kw)
        f = integrands.Normal(0.5, 0.25, d)
    model = NF._model
    model.train()
    optim = torch.optim.Adamax(model.parameters(), lr=2e-3, weight_decay=1e-4)       # README.md:42
    maxf = torch.ones((), dtype=torch.double, device=dev)

    def step():                      # what one epoch of _train_variance_forward_seq does between zero_grad() and optimizer.step()
        optim.zero_grad()
        w = torch.rand(G, mbs, d, dtype=torch.double, device=dev)
        loss_sum, t = flow_loss_batched(model, w, f, maxf, False, False)
        loss = loss_sum / G
        loss.backward()
        optim.step()
        return loss

    for _ in range(5):               # allocator growth, code-object loads and the optimizer's lazy state are behind these
        step()
    barrier_sync(ws)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier_sync(ws)
    dt = max_over_ranks(time.perf_counter() - t0, ws, dev)
    _autograd.TRAIN_GRAPH = False
    rate = G * mbs * steps * ws / dt
    tf = 3.0 * fwd_flop * rate / ws / 1e12          # forward + adjoint ~ 3x the forward's algorithmic flop (dX and dW products per layer)
    return {"workload": f"{cfg} optimizer step: draw, forward with history, f(X)*J, loss + bookkeeping, adjoint, Adamax step "
                        f"({G} minibatch{'es as grid.y' if G > 1 else ''} of {mbs})", "mbs": mbs, "minibatches_per_step": G,
            "value": rate, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "loss": float(loss.detach()),
            "roofline": {"bound": "valu_f64" if cfg == "C2" else "mfma_f64", "achieved": round(tf, 3), "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(tf / MFMA_F64_PEAK_TF, 4), "flop_per_sample": 3.0 * fwd_flop,
                         "note": "3 x SURVEY 8(d) forward flop/sample x samples/s of the whole step"}}


def cpu_baseline():
    """The reference-equivalent CPU path on the GPU box's host cores: oracle/flow_torch.py -- the PyTorch-CPU fp64
    restatement of the C2 integrate loop, pinned to the reference's golden vectors -- on ALL cores (1 warm-up + median of 5
    iterations of 1e6 samples), the NumPy oracle on ONE core beside it; for metric part (ii) oracle/rambo_torch.py -- the PyTorch-CPU
    restatement of generateKinematics_batch, same thread count, 2 -> 4 massive at the metric's 1e6 points -- and the NumPy RAMBO oracle
    on one core (2e5 points)."""
    from oracle import flow as oflow
    from oracle import flow_torch as otorch
    from oracle import rambo as orambo
    from oracle import rambo_torch as orambo_t
    from nf_amd.normalizing_flows.layers.coupling_cells import PWQuad
    torch.manual_seed(0)
    cells = torch.nn.Sequential(PWQuad(2, 1, 4, [3] * 3), PWQuad(2, 1, 4, [3] * 3)).double()
    params = {k: v.detach().cpu().numpy() for k, v in cells.state_dict().items()}
    steps = oflow.plan_pwquad(2, 2, 4)
    host = os.cpu_count() or 1
    tparams = otorch.to_torch(params)
    gen = torch.Generator().manual_seed(0)

    def one(n):
        w = [torch.rand(n, 2, dtype=torch.float32, generator=gen)]
        t0 = time.perf_counter()
        otorch.integrate(otorch.camel, tparams, steps, w)
        return time.perf_counter() - t0

    # thread count: every host core is the ask, but a GPU box grants a job a CPU share (16 cores per GPU on this
    # pool) and 256 torch threads on it ran 700x SLOWER than one NumPy core -- so the count is calibrated on a 2e5-sample
    # iteration over {share candidates .. all cores} and the fastest is used and stated
    cands = sorted({c for c in (8, 16, 32, 64, host) if c <= host})
    if os.environ.get("NIS_CPU_THREADS"):
        cands = [int(os.environ["NIS_CPU_THREADS"])]
    best = None
    for c in cands:
        torch.set_num_threads(c)
        one(20_000)
        t = one(200_000)
        if best is None or t < best[0]:
            best = (t, c)
        if t > 5.0:
            break                                       # oversubscribed beyond use: larger counts only get worse
    cores = best[1]
    torch.set_num_threads(cores)
    times = [one(BATCH) for _ in range(6)]
    t_all = float(np.median(times[1:]))
    rng = np.random.default_rng(0)
    n_it = 3
    draws = [rng.random((BATCH, 2), dtype=np.float32) for _ in range(n_it)]
    t0 = time.perf_counter()
    sig, sig_err, *_ = oflow.integrate(oflow.camel, params, steps, draws)
    dt1 = time.perf_counter() - t0
    n_rb = 200_000
    r = rng.random((n_rb, 8))
    t1 = time.perf_counter()
    orambo.generate_kinematics([100.] * 2, [100.] * 4, 1000., r, 0, 0, -1)
    dtr = time.perf_counter() - t1
    # metric part (ii) on all cores: the torch-CPU restatement at the metric's batch (README.md:53-58 workload), 1 warm-up + median of 3
    rt = torch.rand(BATCH, 8, dtype=torch.double, generator=gen)
    orambo_t.generate_kinematics([100.] * 2, [100.] * 4, 1000., rt[:100_000])
    tr = []
    for _ in range(3):
        t1 = time.perf_counter()
        orambo_t.generate_kinematics([100.] * 2, [100.] * 4, 1000., rt)
        tr.append(time.perf_counter() - t1)
    t_rb = float(np.median(tr))
    return {"value": BATCH / t_all, "unit": "samples/s", "cores": cores, "kind": "port",     # cores = threads used; host_cpus = os.cpu_count()
            "sample": f"oracle.flow_torch.integrate (PyTorch CPU float64, torch.set_num_threads({cores}), fastest of {cands} on this host): C2 model, "
                      f"1e6 samples per iteration, 1 warm-up + median of 5 ({t_all * 1e3:.0f} ms per iteration)",
            "one_core": {"value": n_it * BATCH / dt1, "unit": "samples/s", "cores": 1,
                         "sample": f"oracle.flow.integrate (NumPy float64): {n_it} iterations x 1e6 samples ({dt1:.1f} s)",
                         "integral": float(sig), "sigma": float(sig_err)},
            "rambo": {"value": BATCH / t_rb, "unit": "samples/s", "cores": cores, "kind": "port",
                      "sample": f"oracle.rambo_torch.generate_kinematics (PyTorch CPU float64, {cores} threads): 2->4 massive, E_cm 1000, 1e6 points "
                                f"per call, 1 warm-up + median of 3 ({t_rb:.2f} s per call)",
                      "one_core": {"value": n_rb / dtr, "unit": "samples/s", "cores": 1,
                                   "sample": f"oracle.rambo 2->4 massive (NumPy), {n_rb} points ({dtr:.1f} s)"}},
            "host_cpus": host}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 200 warm-up steps (~7 ms) bring the GPU to its sustained clocks and 1000 timed steps (~30 ms) keep start / stop
    # costs below 1 %; the driver's `--steps 20 --warmup 5` times ONE launch set right after a cold start
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--roofline-kernel", default="apply", choices=["apply", "stats", "ingest"])
    ap.add_argument("--group", type=int, default=20, help="steps fused into one launch set (independent BN batches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C3 / C4 / C5 legs")
    args = ap.parse_args()
    # stdout carries ONE JSON line: file descriptor 1 points at stderr until that line is printed, so that nothing a native library
    # writes there (gloo's "[Gloo] Rank 0 is connected ..." in rehearsals, RCCL / ROCr notices) can get in front of it
    sys.stdout.flush()
    fd_out = os.dup(1)
    os.dup2(2, 1)
    rank, ws, local = init_from_env("nccl")
    if ws != args.gpus and ws > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ws}")
    _lib.require_gpu()
    local = local % max(1, torch.cuda.device_count())      # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):           # the reference API prints ("Adjusted # coupling cells", "Invalid boost"): stdout carries ONE line
        flow = bench_flow(args, rank, ws, dev)
        rambo = bench_rambo(args, rank, ws, dev)
        configs = None if args.no_configs else {name: bench_config(name, rank, ws, dev) for name in CONFIGS}
        train = None if args.no_configs else {name: bench_train_step(name, rank, ws, dev) for name in TRAIN_CASES}
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
        "config": {"workload": "C2: 2-D camel, PWQuad 2 cells, n_bins 4, NN [3,3,3], train-mode BN, draw inside the step (device Philox, float32 "
                               "lattice), fused camel*J + moments",
                   "batch_per_gpu": BATCH, "global_batch": BATCH * ws, "parallelism": f"sample-shard x{ws}",
                   "collective": "one all-reduce (SUM) of the zero-padded (world, steps, 4) moment records + one merge launch (nf_amd/parallel.py: allreduce_moments)", "steps_per_launch_set": flow["G"], "launch_set": ("hipGraph replay" if flow["graph"] else "eager launches"),
                   "backend": (dist.get_backend() if ws > 1 else "none"),
                   "rccl_ranks": (dist.get_world_size() if ws > 1 and dist.get_backend() == "nccl" else 0 if ws > 1 else 1),
                   "samples_per_rank": BATCH * args.steps},
        "integral": {"value": flow["sig"], "sigma": flow["sig_err"], "sigma_mc": flow["sig_mc"], "truth": 0.2323235706,
                     "note": "sigma = the reference's formula (manager.py:403); sigma_mc = 1/sqrt(sum_i N/var_i)"},
        "roofline": flow["roofline"],
        "roofline_hbm": flow["roofline_hbm"],
        "roofline_issue": flow["roofline_issue"],
        "rambo": rambo,
        "configs": configs,
        "train_step": train,
    }
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        with contextlib.redirect_stdout(sys.stderr):
            line["cpu_baseline"] = cpu_baseline()
    sys.stdout.flush()
    os.dup2(fd_out, 1)
    os.close(fd_out)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if ws > 1:
        os.dup2(2, 1)                                      # (teardown messages of the process group: not on stdout either)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
