"""Debug aid: per-step max |HIP - oracle| for SolverFeatherstone on the quadruped scene (run on a GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import newton_amd as nt  # noqa: E402
from oracle_bridge import Oracle, OracleState  # noqa: E402
from scenes import quadruped_scene  # noqa: E402
from test_gpu_parity_xpbd import _lower_quadrupeds  # noqa: E402

model = quadruped_scene(int(sys.argv[2]) if len(sys.argv) > 2 else 3, device="cuda:0")
_lower_quadrupeds(nt, model, 0.22)
o = Oracle(model)
s0, s1 = model.state(), model.state()
pipe = nt.CollisionPipeline(model)
contacts = pipe.contacts()
solver = nt.solvers.SolverFeatherstone(model)
os0, os1 = OracleState(model), OracleState(model)
oc, c = o.contacts(), o.control()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    s0.clear_forces()
    pipe.collide(s0, contacts)
    solver.step(s0, s1, None, contacts, 1e-3)
    s0, s1 = s1, s0
    os0.body_f[:] = 0
    o.collide(os0.body_q, oc)
    o.featherstone_step(os0, os1, c, oc, 1e-3)
    os0, os1 = os1, os0
    dq = np.abs(s0.joint_q.cpu().numpy() - os0.joint_q)
    dqd = np.abs(s0.joint_qd.cpu().numpy() - os0.joint_qd)
    db = np.abs(s0.body_q.cpu().numpy() - os0.body_q)
    print(i, "contacts", int(oc.count[0]), int(contacts.rigid_contact_count.cpu().numpy()[0]), "dq %.2e@%d dqd %.2e@%d dbody %.2e" % (dq.max(), dq.argmax(), dqd.max(), dqd.argmax(), db.max()))
