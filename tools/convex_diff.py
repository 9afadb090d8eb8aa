"""Debug aid: per convex known-answer case, max |HIP - oracle| over the exported contact arrays (run on a GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import newton_amd as nt  # noqa: E402
from oracle_bridge import Oracle  # noqa: E402
from pair_scenes import CONVEX_CASES, pair_model  # noqa: E402

for name in sorted(CONVEX_CASES):
    model = pair_model(CONVEX_CASES[name], device="cuda:0")
    o = Oracle(model)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    pipe.collide(model.state(), contacts)
    oc = o.contacts()
    o.collide(model.body_q, oc)
    n = int(oc.count[0])
    ng = int(contacts.rigid_contact_count.cpu().numpy()[0])
    worst = 0.0
    if n == ng:
        for f in ("point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
            g = getattr(contacts, "rigid_contact_" + f).cpu().numpy()[:n]
            w = getattr(oc, f)[:n]
            if n:
                worst = max(worst, float(np.max(np.abs(g - w))))
    print(f"{name:28s} oracle={n} hip={ng} max_abs_diff={worst:.3e}")
