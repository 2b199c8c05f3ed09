import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package()
import torch
torch.cuda.set_stream(torch.cuda.Stream(0))      # not the legacy null stream: the engine's graph replay needs a capturable stream (as bench.py)
eng = importlib.import_module('rustronomy_watershed_amd.device').DeviceEngine(0)
for S in (128, 256, 512, 1024, 1536, 2048, 3072, 4000):
    res = []
    for seed in (1, 2, 3):
        img = eng.random_field(S, S, seed)
        seeds = eng.find_local_minima(img)
        out = torch.empty((S, S), dtype=torch.int32, device=img.device)
        for i in range(3):
            eng.segment(img, seeds, out=out)
        torch.cuda.synchronize()
        st = eng.ctx.stats()
        res.append((st["relax_passes"], st["graph_launches"]))
    print(S, res, flush=True)
