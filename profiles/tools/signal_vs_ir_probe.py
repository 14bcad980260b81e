"""How much of the forward-transform launch is the signal windows?  cfg3 (moving streams: every window through the enveloped
general path) and cfg2, al_signal_spectra and al_ir_spectra timed as separate launches beside the merged al_forward_spectra."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from audiblelight_amd import engine, plan as planning, synthetic

r = engine.Renderer()
for cfg in sys.argv[1:] or ["cfg3", "cfg2"]:
    sc = synthetic.make_scene(cfg, torch_device="cuda")
    pl = planning.plan_batch(sc.specs, sc.n_capsules, sc.ir_len, sc.sr)
    c, n, l = sc.ir_shape
    batch = r.prepare(pl, sc.sources(), sc.irs_dev, ir_strides=(n * l, l))
    for _ in range(2):
        batch.run()
    torch.cuda.synchronize()
    out = {}
    for name in ("al_forward_spectra", "al_signal_spectra", "al_ir_spectra"):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            batch.run_stage(name)
        b.record()
        torch.cuda.synchronize()
        out[name] = round(a.elapsed_time(b) / 5, 4)
    print(cfg, "log2_block", pl.log2_block, "signal windows", pl.xspec_blocks, "IR partitions", pl.hspec_blocks, out, flush=True)
    del batch, sc
    torch.cuda.empty_cache()
