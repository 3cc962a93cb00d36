import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from mvp_pytorch_amd import hip, modeling, train, dp
from mvp_pytorch_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
dims = dict(B=256, T=70, P=5, G=20, R=50)
torch.manual_seed(1234)
model = modeling.BiBertImgForPreTraining(modeling.make_config(bench.BASE_CFG)).to(dev).train()
opt, sched = train.build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, t_total=100000)
sync = dp.GradSync(model)
b = synthetic_batch(dims, bench.BASE_CFG, 1234, device=dev)
for _ in range(3):
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync, max_grad_norm=10.0)
torch.cuda.synchronize()
orig = hip.tap_rows_bwd
def dbg(taps, rows, rows2, H):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(taps, rows, rows2, H)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) * 1e6
    R = rows + rows2
    allidx = torch.cat([i for _, i in taps]).long()
    v = allidx[(allidx >= 0) & (allidx < R)]
    cnt = torch.bincount(v, minlength=R)
    print("tap_rows_bwd rows %d + %d, taps %s: valid entries %d, max per row %d, rows with >3: %d, >64: %d   %.0f us (synced)" % (
        rows, rows2, [(tuple(g.shape), str(g.dtype)[6:]) for g, _ in taps], v.numel(), int(cnt.max()), int((cnt > 3).sum()), int((cnt > 64).sum()), us), flush=True)
    return out
hip.tap_rows_bwd = dbg
train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync, max_grad_norm=10.0)
torch.cuda.synchronize()
