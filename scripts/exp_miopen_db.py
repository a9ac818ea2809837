"""Experiment: persist MIOpen's user find-db in the repo so cudnn.benchmark picks are instant."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
db = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/miopen_db")
os.makedirs(db, exist_ok=True)
os.environ["MIOPEN_USER_DB_PATH"] = db
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(db, "cache")
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
torch.backends.cudnn.benchmark = True
t0 = time.time()
dev = torch.device("cuda:0")
model = bench.build_model(300, dev)
opt = bench.make_optimizer(model)
np.random.seed(3)
batch = bench.synth_batch(4, 1000, dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(*batch)
    bench.total_cost(out).backward()
    opt.step()
step(); torch.cuda.synchronize()
print("first step (incl. MIOpen find) %.1f s" % (time.time() - t0))
for _ in range(3): step()
torch.cuda.synchronize(); t = time.time()
for _ in range(8): step()
torch.cuda.synchronize(); dt = (time.time() - t) / 8
print("benchmark mode with db %s: %.1f ms/step, %.2f pairs/s" % (db, dt * 1e3, 4 / dt))
os.system("du -sh %s; ls -la %s | head" % (db, db))
