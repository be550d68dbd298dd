import numpy as np, torch, sys, math
sys.path.insert(0,'/root/repo')
from kinpoly_amd import sim as kpsim
from kinpoly_amd.context import get_qvel_fd_batch, quat_mul, quat_inv
from oracle import np_oracle as O
rng = np.random.default_rng(5)
n, dt = 300, 1.0 / 30.0
qpos = rng.normal(0, 0.4, (n, 76)); qpos[:, 2] += 0.9
qpos[:, 3:7] = rng.normal(0, 1, (n, 4)); qpos[:, 3:7] /= np.linalg.norm(qpos[:, 3:7], axis=1, keepdims=True)
act = rng.normal(0, 0.5, (n, 80))
act[:100, 77:80] *= 10.0 ** rng.uniform(-3, -1, (100, 1)); act[-1,77:80]=0
nxt, qv = kpsim.kin_advance(torch.tensor(qpos, dtype=torch.float32, device="cuda"), torch.tensor(act, dtype=torch.float32, device="cuda"), dt)
torch.cuda.synchronize()
q32 = torch.tensor(qpos, dtype=torch.float32).double().numpy(); a32 = torch.tensor(act, dtype=torch.float32).double().numpy()
want = np.stack([O.step_ar(q32[i], a32[i], dt) for i in range(n)])
want[:, 3:7] /= np.linalg.norm(want[:, 3:7], axis=1, keepdims=True)
q32n = q32.copy(); q32n[:, 3:7] /= np.linalg.norm(q32n[:, 3:7], axis=1, keepdims=True)
wantv = get_qvel_fd_batch(torch.tensor(q32n), torch.tensor(want), dt).numpy()
got = qv.cpu().numpy()
d=np.abs(got[:-1,3:6]-wantv[:-1,3:6]).max(1)
for i in np.argsort(d)[-5:]:
    qrel = quat_mul(torch.tensor(want[i:i+1,3:7]), quat_inv(torch.tensor(q32n[i:i+1,3:7])))
    print(i, d[i], "got", got[i,3:6], "want", wantv[i,3:6], "|omega| action", np.linalg.norm(a32[i,77:80]), "1-w", 1-qrel[0,0].item())
