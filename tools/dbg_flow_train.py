import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tests/golden"))
from test_hip_seq_train import _random_flow, _rel, _away_from_the_kink
from synth import seeded_randn
from oracle import behavior_oracle as B
lr = 4.5e-7 * 64
nb, steps = int(sys.argv[1]), int(sys.argv[2])
flow, sd = _random_flow(1024, 2048, 2, nb, 7)
if len(sys.argv) > 3:
    sd = _away_from_the_kink(sd, 2)
    flow.load_state_dict(sd)
eng = flow.flow.train_engine(lr=lr, betas=(0.5, 0.9), weight_decay=0.0)
eng.graph.enabled = False
ref = {k: v.clone() for k, v in sd.items()}
opt = B.flow_optimizer(ref, lr, 0.0)
for it in range(steps):
    bs = seeded_randn(f"w.b{it}", (64, 1024), 7)
    print(eng.train_step(bs.cuda(), torch.zeros(64, 1024, device="cuda")).tolist())
    print(B.flow_train_step(ref, opt, bs))
names = B.flow_parameters(ref)
mine = eng.optimizer_state_dict()["state"]
theirs = opt.state_dict()["state"]
for i, n in enumerate(names):
    a, b = mine[i]["exp_avg"].double().cpu(), theirs[i]["exp_avg"].double()
    d = ((a - b).abs() / b.abs().max()).flatten()
    if d.max() > 2e-5:
        ds = d.sort().values
        k = d.numel()
        rows = (a - b).abs().reshape(a.shape[0], -1).max(dim=1).values / b.abs().max()
        print(f"{n:56s} q50 {ds[k // 2]:.1e} q99 {ds[int(k * .99)]:.1e} q999 {ds[int(k * .999)]:.1e} max {ds[-1]:.1e}; rows with err > 1e-4: {int((rows > 1e-4).sum())} of {rows.numel()}")
print("done")
