"""Accuracy sweep of the eigensolver over awkward sizes (panel / tail boundaries): python scripts/eigh_sweep.py [n ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from janusx_amd import pipeline as jp   # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [256, 257, 258, 300, 319, 320, 321, 383, 384, 385, 447, 448, 449, 511, 513, 1000, 1217]
bad = 0
for n in sizes:
    rng = np.random.default_rng(n)
    z = rng.standard_normal((n, n + 11))
    k = z @ z.T / z.shape[1]
    w, u = jp.eigh_from_grm(torch.from_numpy(k).cuda(), ridge=0.0)
    wh, uh = w.cpu().numpy(), u.cpu().numpy()
    wref = np.linalg.eigvalsh(k)
    sc = np.abs(wref).max()
    ev = np.abs(np.sort(wh) - wref).max() / sc
    rs = np.abs(k @ uh.T - uh.T * wh[None, :]).max() / sc
    ob = np.abs(uh @ uh.T - np.eye(n)).max()
    ok = ev < 1e-12 and rs < 1e-12 and ob < 1e-12
    bad += not ok
    print(f"n={n:5d} eval_err={ev:.2e} resid={rs:.2e} orth={ob:.2e} {'ok' if ok else 'FAIL'}")
sys.exit(1 if bad else 0)
