import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from tests.test_search_gpu import _engine, _run_engine_search, _oracle, FENS
from tests.fake_net import FakeNet
fen = FENS[1]
for vl, dirichlet in [(False, False), (True, False), (False, True)]:
    for sims in [1, 2, 4, 8, 9, 12, 16, 24, 32, 48, 64, 96]:
        e, m = _engine(1, 8, sims, vl_active=vl)
        e.search_begin(0, fen, sims, dirichlet, 101)
        r = _run_engine_search(e, FakeNet(seed=3, sharp=8.0), 1)[0]
        o, b, vc, pi, rq = _oracle(fen, 101, sims, 8, FakeNet(seed=3, sharp=8.0), m, vl, dirichlet)
        kids = list(o._last_root.children.values())
        on = [c.n for c in kids]
        ok = r["n"].tolist() == on
        print(vl, dirichlet, sims, ok, "root_n", r["root_n"], o._last_root.n, "evals", o.evals, flush=True)
        if not ok:
            print(" eng", r["n"].tolist()); print(" ora", on)
            print(" prior diff", np.abs(r["prior"] - np.array([c.prior for c in kids])).max())
            break
        e.close()
