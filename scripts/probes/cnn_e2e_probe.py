import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dsac-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "envs")):
    sys.path.insert(0, p)
import plugin
from test_hip_cnn_parity import cnn_kwargs
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6001
kw = cnn_kwargs((3, 96, 96), 3, "type_2", 64, env_id="synth_blob", sample_batch_size=20, reward_scale=1,
                buffer_warm_size=400, buffer_max_size=50000, max_iteration=iters, log_save_interval=500,
                apprfunc_save_interval=100000, eval_interval=500, num_eval_episode=5, ini_network_dir=None,
                save_folder=None, seed=2024, sample_interval=1, strict_rng=False)
torch.manual_seed(kw["seed"]); np.random.seed(kw["seed"])
alg = plugin.create_alg(**kw); sampler = plugin.create_sampler(**kw); buf = plugin.create_buffer(**kw)
ev = plugin.create_evaluator(**kw); tr = plugin.create_trainer(alg, sampler, buf, ev, **kw)
tars = []; orig = ev.run_evaluation
def hook(it):
    t = orig(it); tars.append(t)
    st = alg.engine.read_stats()
    print(it, round(t, 2), "alpha %.3f q1 %.3f critic_loss %.3f" % (st["DSAC2/alpha-RL iter"], st["DSAC2/critic_avg_q1-RL iter"], st["Loss/Critic loss-RL iter"]), flush=True)
    return t
ev.run_evaluation = hook
t0 = time.time(); tr.train(); print("wall", time.time() - t0)
