import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ait_ref
from oracle.digest import seeded
from ait_amd.system import Transformer
sd = ait_ref.make_ait_state_dict(seed=3)
t = Transformer(d_k=64, d_v=64, d_model=512, d_word_vec=512, d_inner=2048, n_position=64, n_layers=1, n_head=8, dropout=0.1)
t.load_state_dict(sd); t = t.cuda().eval()
xp = torch.from_numpy(seeded(301, (6, 1024, 7, 7))); xq = torch.from_numpy(seeded(302, (2, 1024, 8, 8)))
cot = torch.from_numpy(seeded(303, (6, 1024, 8, 8)))
a = xp.clone().requires_grad_(True); b = xq.clone().requires_grad_(True)
sdr = {k: v.clone().requires_grad_(True) if 'pos_table' not in k else v for k, v in sd.items()}
y = ait_ref.transformer_forward(sdr, a, b)
ga, gb = torch.autograd.grad(y, [a, b], cot)
A = xp.cuda().requires_grad_(True); B = xq.cuda().requires_grad_(True)
Y = t(x_props=A, x_query=B)
GA, GB = torch.autograd.grad(Y, [A, B], cot.cuda())
err = (GA.cpu() - ga).abs()
print("max err", err.max().item(), "ref max", ga.abs().max().item())
print("per-proposal max err", err.amax(dim=(1, 2, 3)))
print("per-token max err", err.view(6, 1024, 49).amax(dim=(0, 1)))
ec = err.view(6, 1024, 49).amax(dim=(0, 2))
print("per-channel: n bad", (ec > 1e-4).sum().item(), "first bad", torch.nonzero(ec > 1e-4)[:10].flatten().tolist())
print("---- with param grads requested")
params = dict(t.named_parameters())
A = xp.cuda().requires_grad_(True); B = xq.cuda().requires_grad_(True)
Y = t(x_props=A, x_query=B)
gs = torch.autograd.grad(Y, [A, B] + list(params.values()), cot.cuda())
err = (gs[0].cpu() - ga).abs()
print("max err", err.max().item(), gs[0].stride(), gs[0].shape)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/g3_transformer.npz"))
samp = g["t23/g_x_props/sample"]
flat = ga.numpy().reshape(-1); stride = flat.size // 4096
print("oracle vs golden sample max", np.abs(flat[::stride][:4096] - samp).max())
f2 = gs[0].detach().cpu().numpy().reshape(-1)
print("hip vs golden sample max", np.abs(f2[::stride][:4096] - samp).max())
