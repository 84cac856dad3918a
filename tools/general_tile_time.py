"""general_matmul_*: one wavefront per series with lanes over rows (c2_general_tile.hip) vs the earlier kernels
(two-phase for 1-2 right-hand sides, lanes over the right-hand sides from 3).  Usage: general_tile_time.py [B [J]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
J = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = 4096
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
t1 = (t + 0.013).contiguous()
def timed(f, reps=7):
    return synth.timed_steady(f, reps=reps)   # (steady clock: profiles/r05_clock_ramp.md)
for nrhs in (1, 2, 4, 8):
    Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
    Z = torch.empty((B, N, nrhs), dtype=torch.float64, device=dev)
    F = torch.empty((B, N, J, nrhs), dtype=torch.float64, device=dev)
    for name in ("general_matmul_lower", "general_matmul_upper"):
        for wf in (False, True):
            res = {}
            for mode in ("0", "1"):
                os.environ["C2_GENERAL_TILE"] = mode
                f = getattr(ops, name)
                if wf:
                    F.fill_(-7.0)
                    call = lambda: f(t1, t, c, U, V, Y, Z=Z, F=F, zero_z=True)
                else:
                    call = lambda: f(t1, t, c, U, V, Y, Z=Z, zero_z=True)
                ms = timed(call)
                res[mode] = (ms, Z.clone(), F.clone() if wf else None)
            alg = B * 8.0 * N * ((1 + J + nrhs) * 2 + (J * nrhs if wf else 0))
            dz = float((res["0"][1] - res["1"][1]).abs().max() / res["0"][1].abs().max())
            df = float((res["0"][2] - res["1"][2]).abs().max() / res["0"][2].abs().max()) if wf else 0.0
            print("%s%s nrhs=%d: before %.2f ms, row tiles %.2f ms (frac %.3f), rel diff Z %.1e F %.1e" % (
                name, " +F" if wf else "", nrhs, res["0"][0], res["1"][0], alg / res["1"][0] / 8e9, dz, df), flush=True)

# independent random grids (outputs straddle the state tiles) and small batches of long series
torch.manual_seed(1)
for (B2, N2) in ((B, 4096), (1, 100_000), (64, 100_000)):
    t2g, c2g, a2g, U2g, V2g, y2g = synth.device_batch_fast(1, B2, N2, J, dev)
    span = (t2g[:, -1:] - t2g[:, :1])
    t1g = (t2g[:, :1] + span * torch.rand((B2, N2), dtype=torch.float64, device=dev)).sort(dim=1).values.contiguous()
    Y = torch.randn((B2, N2, 1), dtype=torch.float64, device=dev)
    Z = torch.empty((B2, N2, 1), dtype=torch.float64, device=dev)
    res = {}
    for mode in ("0", "1"):
        os.environ["C2_GENERAL_TILE"] = mode
        res[mode] = (timed(lambda: ops.general_matmul_lower(t1g, t2g, c2g, U2g, V2g, Y, Z=Z, zero_z=True)), Z.clone())
    print("random grids B=%d N=M=%d nrhs=1: before %.3f ms, row tiles %.3f ms, rel diff %.1e" % (
        B2, N2, res["0"][0], res["1"][0], float((res["0"][1] - res["1"][1]).abs().max() / res["0"][1].abs().max())), flush=True)
