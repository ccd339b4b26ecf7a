#!/bin/bash
# GPU box: CTC parity (both shapes), cfg2 step through the C ABI and the operator, the repair regime
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "ctc" 2>&1 | tail -3
WFL_CTC_MITM_WAVES=8 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "ctc" 2>&1 | tail -2
python bench.py --config cfg2 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('operator %.0f utt/s %.4f ms; kernels %s; abi %.4f ms; fresh %.4f ms' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['abi_kernels_only']['ms_per_step'], r['fresh_targets']['ms_per_step']))"
python scripts/spread_probe.py 2>&1 | grep randn | head -5
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from gtn_applications_amd import engine as E
B, T, C, L = 128, 1000, 100, 44
g = torch.Generator().manual_seed(0)
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
for s in (1.0, 1.5, 3.0):
    x = (s * torch.randn(B, T, C, generator=g)).cuda()
    tg = E.targets_on_device(targets, x.device)
    scale, _, coef = E.loss_factors(tg, "none")
    dx = torch.empty_like(x)
    for _ in range(5): ws, nll = E.ctc_forward_backward(x, tg, C - 1, coef, None, dx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ws, nll = E.ctc_forward_backward(x, tg, C - 1, coef, None, dx)
    torch.cuda.synchronize()
    print("scale %.1f: %.1f us per step, repaired %d" % (s, (time.perf_counter() - t0) / 50 * 1e6, E.ctc_pipeline_repaired(ws, B, T, tg.max_len)))
PY
