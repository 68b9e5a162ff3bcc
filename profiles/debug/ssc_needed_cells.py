"""How many (representative row, lattice node) cells does a flux request actually query?  The reference builds a cell's SSC spectrum
lazily, on the first query (inverse-compton.h:614-620); the engine builds a table for every cell.  A cell (theta row j, node k) is
queried if for SOME phi the interval before or after node k holds a requested time.  usage: python profiles/debug/ssc_needed_cells.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "profiles"))
import vegasafterglow_amd as va
from ssc_ensemble import c3_batch, c5_batch

t_req = np.logspace(2, 8, 100)
for name, prms in (("C3", c3_batch(8)), ("C5", c5_batch(8))):
    for i in (0, 3, 7):
        m = va.Model.from_params(prms[i])
        d = m.details(t_req.min(), t_req.max())
        t_obs = d.fwd.t_obs  # [n_phi_eff][n_theta][n_t]
        nphi, nth, K = t_obs.shape
        idx = np.searchsorted(t_req, t_obs)           # number of requested times below each node
        has = np.zeros((nphi, nth, K), bool)            # interval [k, k+1) holds a requested time
        has[:, :, :-1] = idx[:, :, 1:] > idx[:, :, :-1]
        need = has.copy()
        need[:, :, 1:] |= has[:, :, :-1]                # node k closes interval k-1
        cell_need = need.any(axis=0)                    # over phi (cells are shared by the phi rows of a theta row)
        # the conservative device-side test: some phi has t_obs[k-1] <= t_max_req and t_obs[k+1] >= t_min_req
        lo = np.concatenate([t_obs[:, :, :1], t_obs[:, :, :-1]], axis=2)
        hi = np.concatenate([t_obs[:, :, 1:], t_obs[:, :, -1:]], axis=2)
        cons = ((lo <= t_req.max()) & (hi >= t_req.min())).any(axis=0)
        print(f"{name} member {i}: grid {t_obs.shape}  cells queried exactly {cell_need.mean():.3f}  by the range test {cons.mean():.3f}"
              f"  (row-node pairs queried {need.mean():.3f})", flush=True)
