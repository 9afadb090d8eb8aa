"""Population statistics of the staged mesh-SDF narrow phase on the sdf_bin workload (measurement tool): runnable pairs, pairs
with culling survivors, survivors / valid contacts / exported rows per pair."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench  # noqa: E402

import newton_amd as nt  # noqa: E402


def main():
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    model = bench.build_shard("sdf_bin", envs, 0, 1, "cuda:0")
    pipe = nt.CollisionPipeline(model, broad_phase="sap")
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    s0, s1, ctrl = model.state(), model.state(), model.control()
    dt = bench.WORKLOADS["sdf_bin"]["dt"]
    for _ in range(frames * bench.SUBSTEPS):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, ctrl, contacts, dt)
        s0, s1 = s1, s0
    pipe.collide(s0, contacts)
    torch.cuda.synchronize()
    leg = pipe._sdf_leg
    live = int(leg.pair_prefix[-1].item())
    blk = leg.hit_blk.cpu().numpy().reshape(-1, 4)
    surv = blk[:, 1] + blk[:, 3]
    fp = leg.hit_fp.cpu().numpy()
    valid = np.zeros(len(blk), dtype=np.int64)
    for col in (0, 2):
        off, cnt = blk[:, col], blk[:, col + 1]
        for p in np.flatnonzero(cnt)[:200000]:
            valid[p] += int((fp[off[p]:off[p] + cnt[p]] >= 0).sum())
    rows = leg.blk.cpu().numpy().reshape(-1, 2)[:, 1]
    with_surv = surv > 0
    out = {
        "envs": envs, "live_pairs": live, "runnable_pairs": int(leg.hit_count[1].item()), "pairs_with_survivors": int(with_surv.sum()),
        "survivors": int(surv.sum()), "pairs_with_rows": int((rows > 0).sum()), "rows": int(rows.sum()),
        "survivors_per_pair_hist": np.bincount(np.minimum(surv[with_surv], 64), minlength=65).tolist(),
        "rows_per_pair_hist": np.bincount(np.minimum(rows[rows > 0], 32), minlength=33).tolist(),
        "valid_per_pair_hist_sampled": np.bincount(np.minimum(valid[with_surv][:200000], 64), minlength=65).tolist(),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
