"""Experiment driver with the interface of the reference's `nisrep/utils/experiment_mg.py:12-124` (`pro(para)`): train a
PWQuad flow on the integrand `para["f"]`, report cross section, final variance and unweighting efficiency to
`<logdir>/<id>/log.txt` and to the result queue `para["q"]`.  One call = one hyper-parameter point; the reference runs
several of them as OS processes, one per device (`para["dev"]`).

Same `para` keys (n_flow, n_bins, NN_width, NN_length, dev, lr, weight_decay, var_n, batch_size, pt, q, id, f, logdir,
log), same queue tuple and the same lines in log.txt; everything numeric runs through the HIP path of nf_amd (the
variance / mean / max of f(X)*J come from one `nis_moments` launch instead of three torch reductions)."""
import datetime
import os
import pprint

import torch

from ..flow import moments as hip_moments
from ..normalizing_flows.manager import PWQuadManager

epoch_length = 800                       # training epochs per point (module constant, experiment_mg.py:10)
GEV2_TO_PB = 2.56819e-9                  # conversion used for the report lines (experiment_mg.py:63-64)


def pro(para):
    n_flow, n_bins = para["n_flow"], para["n_bins"]
    width, depth = para["NN_width"], para["NN_length"]
    dev, lr, wd = para["dev"], para["lr"], para["weight_decay"]
    var_n, batch_size, pt = int(para["var_n"]), para["batch_size"], para["pt"]
    queue, point_id, f, log = para["q"], para["id"], para["f"], para["log"]
    logdir = para["logdir"] + "/" + str(point_id)
    os.makedirs(logdir, exist_ok=True)
    t_start = datetime.datetime.utcnow()
    with open(logdir + "/log.txt", "w+") as out:
        out.write("Started \n")
        out.write(f"Batchsize: {batch_size} n_bins: {n_bins} NN_length: {depth} NN_width: {width}\n")
        out.write(f"LR: {lr} weight decay: {wd} preburn_time: {pt}\n")
        out.write("-----------\n")
        NF = PWQuadManager(n_flow=n_flow)
        out.write("Create Model \n")
        out.write(f"Device: {dev}\n")
        NF.create_model(n_cells=2, n_bins=n_bins, NN=[width] * depth, dev=dev)      # cell count adapts to n_flow
        out.write("Model Created \n")
        optim = torch.optim.Adamax(NF._model.parameters(), lr=lr, weight_decay=wd)
        out.write("Start training \n")
        sig, sig_err = NF._train_variance_forward_seq(f, optim, log, logdir, batch_size, epoch_length, pretty_progressbar=False,
                                                      save_best=True, run=None, dev=dev, integrate=True,
                                                      mini_batch_size=batch_size, preburn_time=pt)
        out.write("End training \n")
        t_end = datetime.datetime.utcnow()
        out.write("{0:5E}  +/- {1:3E}  \n".format(sig, sig_err))
        sig, sig_err = sig / GEV2_TO_PB, sig_err / GEV2_TO_PB

        # final variance and unweighting efficiency on var_n fresh points through the best model
        device = NF._device(dev)
        with torch.no_grad():
            w = torch.rand(var_n, n_flow, device=device)                            # float32 draws, as the reference
            XJ = NF.best_model(NF.format_input(w, device))
            mom = hip_moments(f(XJ[:, :-1]), XJ[:, -1]).cpu()
        v_var = float(mom[2] / (mom[0] - 1.0))
        w_mean, w_max = float(mom[1]), float(mom[3])
        seconds = (t_end - t_start).total_seconds()
        queue.put((NF.best_loss.detach().tolist(), None, NF.best_loss_rel.detach().tolist(), NF.best_func_count,
                   NF.varJ.detach().tolist(), NF.DKL.detach().tolist(), NF.best_var, NF.best_epoch, "NIS", seconds, point_id,
                   sig, sig_err, v_var))

        out.write("Final Variance: {0:5E} \n".format(v_var))
        out.write("{0:5E}  +/- {1:3E} pb \n".format(sig, sig_err))
        sig2, sig2_err = NF.integrate(f, 10, var_n, dev)
        out.write("Post training integrate:\n")
        out.write("{0:5E}  +/- {1:3E} pb \n".format(float(sig2) / GEV2_TO_PB, float(sig2_err) / GEV2_TO_PB))
        out.write("Unweighting efficiency: " + str(w_mean / w_max) + "\n")
        out.write("Duration:\n" + str(seconds) + "\n")
        out.write("-----------\n")
        for title, value in (("Initial loss", NF.int_loss.tolist()), ("Best loss", NF.best_loss.tolist()),
                             ("Best loss relative", NF.best_loss_rel.tolist()), ("Evaluations", NF.best_func_count),
                             ("Epoch", NF.best_epoch)):
            out.write(title + "\n" + str(value) + "\n")
        out.write("---------------\n")
        out.write(pprint.pformat({k: v for k, v in para.items() if k not in ("q", "f")}))
