cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_losses.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import torch, time
from triplaneturbo_amd import ops
x = torch.randn(8388608, 3, device="cuda")
for _ in range(5): ops.eikonal_loss(x)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.eikonal_loss(x)
e1.record(); torch.cuda.synchronize()
print("eikonal fwd op: %.1f us per call (incl. slot zeroing + torch wrapper)" % (e0.elapsed_time(e1)/50*1e3))
PY
for r in 1 2; do
bash tools/abn.sh 1 "--steps 100" 2>&1 | cut -c1-200
(cd _r3 && bash tools/abn.sh 1 "--steps 100" 2>&1 | cut -c1-200 | sed "s/^/r3 /")
done
