#!/usr/bin/env python3
"""Large planes on one device: time and defining-equation check (device side) at 16384^2 and 32768^2."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_hip(); ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
eng = dev.DeviceEngine(0)
for n in [int(a) for a in sys.argv[1:]] or [16384]:
    img = eng.random_field(n, n, 1)
    seeds = eng.find_local_minima(img)
    labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
    eng.segment(img, seeds, out=labels)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 3
    for _ in range(K): eng.segment(img, seeds, out=labels)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    # every coloured non-seed pixel carries the label of one of its 4 neighbours; seeds carry index + 1
    lab = labels
    s = seeds.to(torch.int64)
    ok_seeds = bool((lab[s[:, 0], s[:, 1]].to(torch.int64) == torch.arange(1, s.shape[0] + 1, device=lab.device)).all())
    inner = lab[1:-1, 1:-1]
    same = (inner == lab[2:, 1:-1]) | (inner == lab[:-2, 1:-1]) | (inner == lab[1:-1, 2:]) | (inner == lab[1:-1, :-2])
    isseed = torch.zeros_like(lab, dtype=torch.bool); isseed[s[:, 0], s[:, 1]] = True
    ok_nb = bool((same | isseed[1:-1, 1:-1] | (inner == 0)).all())
    print(f"{n}x{n}: {dt*1e3:.3f} ms  {n*n/dt/1e9:.1f} Gpx/s  seeds {seeds.shape[0]}  coloured {int((lab != 0).sum())}  seeds_ok {ok_seeds}  neighbour_ok {ok_nb}  stats {eng.stats()['relax_passes']} passes", flush=True)
    del img, seeds, labels, lab, inner, same, isseed
    torch.cuda.empty_cache()
