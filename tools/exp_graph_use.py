import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package()
import torch
torch.cuda.set_stream(torch.cuda.Stream(0))      # not the legacy null stream: the engine's graph replay needs a capturable stream (as bench.py)
eng = importlib.import_module('rustronomy_watershed_amd.device').DeviceEngine(0)
for S in (2048, 8192):
    img = eng.random_field(S, S, 1)
    seeds = eng.find_local_minima(img)
    out = torch.empty((S, S), dtype=torch.int32, device=img.device)
    for i in range(8):
        eng.segment(img, seeds, out=out)
        torch.cuda.synchronize()
        st = eng.stats()
        print(S, i, st["graph_launches"], st["relax_passes"], st["launches_relax"], flush=True)
