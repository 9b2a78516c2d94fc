"""Print the headline numbers of gpurun_out/r03_bench_{driver,default}.json."""
import json, sys
for tag in ("driver", "default"):
    try:
        d = json.load(open(f"gpurun_out/r03_bench_{tag}.json"))
    except Exception as e:
        print(tag, "missing", e)
        continue
    print(f"[{tag}] C2 {d['value']:.4e} samples/s ({d['ms_per_step']*1e3:.2f} us/step)  rambo {d['rambo']['value']:.4e} ({d['rambo']['roofline']['avg_launch_us']} us, frac {d['rambo']['roofline']['frac']})")
    print("   roofline", d["roofline"]["bound"], d["roofline"]["frac"], "other", d["roofline"].get("other_bound"), "issue", d["roofline_issue"] and d["roofline_issue"]["frac"])
    for k, v in (d.get("configs") or {}).items():
        print(f"   {k}: {v['value']:.4e} samples/s  {v['ms_per_step']:.3f} ms/step  frac {v['roofline']['frac']}  mfma_util {v['roofline']['mfma_util']}")
    for k, v in (d.get("train_step") or {}).items():
        print(f"   train {k}: {v['value']:.4e} samples/s  {v['ms_per_step']:.3f} ms/step  frac {v['roofline']['frac']}")
    cb = d.get("cpu_baseline")
    if cb:
        print("   cpu", f"{cb['value']:.3e}", "cores", cb["cores"], "host_cpus", cb["host_cpus"], "| one core", f"{cb['one_core']['value']:.3e}", "| rambo", f"{cb['rambo']['value']:.3e}")
