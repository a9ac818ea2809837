"""Records MIOpen's user find-db for one bench configuration into a directory (one run with
torch.backends.cudnn.benchmark = True: MIOpen searches every convolution shape it has not seen).  Start from the
committed db so that only NEW shapes are searched:  python scripts/exp_miopen_db.py <dir> [cfg2|cfg3|cfg5]
then copy the grown *.ufdb.txt / *.udb.txt back to ait_amd/miopen_db/."""
import os, shutil, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
db = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/miopen_db")
name = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
os.makedirs(db, exist_ok=True)
src = os.path.join(ROOT, "ait_amd", "miopen_db")
for f in os.listdir(src):
    if not os.path.exists(os.path.join(db, f)):
        shutil.copy(os.path.join(src, f), os.path.join(db, f))
os.environ["MIOPEN_USER_DB_PATH"] = db
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join("/tmp", "miopen_cache_exp")
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from ait_amd import ops
torch.backends.cudnn.benchmark = True
conf = bench.CONFIGS[name]
t0 = time.time()
dev = torch.device("cuda:0")
ops.set_matmul_dtype(conf["dtype"])
model = bench.build_model(conf["proposals"], dev, conf["variant"], conf["layers"])
opt = bench.make_optimizer(model)
np.random.seed(3)
batch = bench.synth_batch(conf["bs"], 1000, dev, max_gt=50 if conf["variant"] == "coco" else 20)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(*batch)
    bench.total_cost(out).backward()
    opt.step()
step(); torch.cuda.synchronize()
print("%s: first step (incl. MIOpen find) %.1f s" % (name, time.time() - t0), flush=True)
for _ in range(3): step()
torch.cuda.synchronize(); t = time.time()
for _ in range(8): step()
torch.cuda.synchronize(); dt = (time.time() - t) / 8
print("%s: benchmark mode with db %s: %.1f ms/step, %.2f pairs/s" % (name, db, dt * 1e3, conf["bs"] / dt))
os.system("wc -l %s/*.txt" % db)
