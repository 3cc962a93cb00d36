# Round 5: compact_scored with its labels requested together and a wave scan (39 us -> ?)
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -q -m gpu -k "compact or packed_training or pretrain_step or finetune" 2>&1 | grep "passed\|failed" | tail -2
python3 - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from mvp_pytorch_amd import hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
labels = torch.randint(0, 30522, (256, 75), generator=g); labels[torch.rand(256, 75, generator=g) < 0.85] = -1
n = int((labels > -1).sum()); labels = labels.to(dev)
pos = torch.randint(0, 40000, (512, 125), generator=g, dtype=torch.int32).to(dev)
for _ in range(3): hip.compact_scored(labels, pos, n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): hip.compact_scored(labels, pos, n)
e1.record(); torch.cuda.synchronize()
print("compact_scored 256 x 75: %.1f us per call (incl. its two output allocations)" % (e0.elapsed_time(e1) / 50 * 1e3))
PY
for r in 1 2; do python3 bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('packed', d['ms_per_step'], d['value'])"; done
