"""Developer probe: 2 gloo ranks on one card (TorchExchange, host staging), handle on torch's
current (NULL) stream like bench.py, vs own stream.  torchrun --nproc-per-node 2 tools/reh_probe.py <use_null>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from bundle_adjustment_solver_amd import scenes
from bundle_adjustment_solver_amd._lib import make_options
from bundle_adjustment_solver_amd.solver import BaProblem
from bundle_adjustment_solver_amd.sharding import TorchExchange
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
use_null = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
pr = scenes.scaled_problem(scenes.config_scene("C3", 0.05))
p = BaProblem(0)
p.set_cameras(pr["cam_intr"], pr["cam_T"]); p.set_poses(pr["pose_T"], pr["pose_fixed"])
p.set_points(pr["pt_X"], pr["pt_fixed"])
p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
p.set_shard(rank, world)
if use_null: p.set_stream(torch.cuda.current_stream().cuda_stream)
p.finalize()
ex = TorchExchange(p, dist, torch.device("cuda", 0), stage_host=True)
mode = sys.argv[2] if len(sys.argv) > 2 else "solve"
opt = make_options(max_iter=6, thr_step=-1, thr_cost=-1)
if mode == "solve":
    rows, _ = p.solve(opt)
else:
    p.lm_begin(opt); p.lm_iterate(2); p.lm_sync(); p.lm_iterate(3); p.lm_sync()
    rows, n, conv, done = p.lm_sync(cap=6)
print("rank", rank, "null" if use_null else "own", mode, " ".join("%d:%.6g" % (r.iteration_status, r.trial_cost) for r in rows), flush=True)
dist.barrier(); dist.destroy_process_group()
