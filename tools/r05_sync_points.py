import os, sys, warnings, collections, traceback
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np, torch
from m2h import synthetic as syn, ops
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
tr = PPOTrainer(near_target_config(sep_update_math="bf16x3"), dev); tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_state_dict(syn.policy_shapes(), 1).items()})
tr.train_cycle(); tr.train_cycle(); torch.cuda.synchronize()
cnt = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "m2h" in f.filename and "site-packages" not in f.filename]
    key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-3:])
    cnt[key] += 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
tr.train_cycle()
torch.cuda.set_sync_debug_mode("default")
for k, v in cnt.most_common(30): print(v, k)
