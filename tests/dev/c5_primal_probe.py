import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import CONFIGS, make_model, synth_batch
from cmf_amd import engine as E
dataset, over, B, off, label = CONFIGS["c5"]
cfg, schema, shape, sd, dens = make_model(torch.device("cuda"), dataset=dataset, overrides=over)
inner = dens.module.density
for Bx in (32, 64):
    x = synth_batch(dataset, shape, Bx, 0, "cuda")
    for thr in (16, 10**9):
        E.GROUPED_PRIMAL_MIN_BATCH = thr
        with torch.no_grad():
            out = inner.elbo(x.clone(), add_reconstruction=True)["elbo"]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): inner.elbo(x.clone(), add_reconstruction=True)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"B={Bx} grouped_min={thr}: {1e3*dt:.2f} ms  elbo[0]={float(out[0]):.4f}", flush=True)
