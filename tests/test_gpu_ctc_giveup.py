"""The meet-in-the-middle CTC launch when a hand-off never arrives (-m gpu): no trap.  A wave that has waited out its
bound raises the launch's status and doubt words and ends; the repair launch behind it recomputes EVERY utterance in
the log domain, so the caller still receives correct losses and gradients, and the give-up is reported (workspace
field WFL_CTC_WS_STATUS, the caller's host_state words).  Forced here with WFL_CTC_MITM_SPIN=1 (one poll) in a
subprocess -- the variable is read once per process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np, torch
from gtn_applications_amd import engine as E
from oracle import recurrences as OR
out = {}
for name, (B, T, C, L) in {"cfg1": (8, 150, 28, 44), "ragged": (5, 333, 40, 0)}.items():
    rs = np.random.RandomState(3)
    x = rs.randn(B, T, C).astype(np.float32)
    targets = [rs.randint(0, C - 1, size=(L or n)).tolist() for n in (7, 1, 30, 12, 3, 9, 2, 5)[:B]]
    want_loss, want_dx = OR.ctc_loss_grad_batched(x, targets, C - 1)
    xd = torch.tensor(x).cuda()
    tg = E.targets_on_device(targets, xd.device)
    scale, _, coef = E.loss_factors(tg, "none")
    dx = torch.full_like(xd, float("nan"))
    ws, nll = E.ctc_forward_backward(xd, tg, C - 1, coef, None, dx)
    torch.cuda.synchronize()
    got_dx = dx.cpu().numpy()
    out[name] = dict(gave_up=bool(E.ctc_pipeline_gave_up(ws, B, T, tg.max_len)), repaired=int(E.ctc_pipeline_repaired(ws, B, T, tg.max_len)),
                     B=B, loss_err=float(np.abs(nll.cpu().numpy() - want_loss).max() / np.abs(want_loss).max()),
                     dx_err=float(np.abs(got_dx - want_dx).max()), finite=bool(np.isfinite(got_dx).all()))
# ... and a process that goes on working afterwards (the context survived): the operator, twice
from gtn_applications_amd.criterions import ctc
xr = torch.tensor(x).cuda().requires_grad_(True)
for _ in range(2):
    xr.grad = None
    loss = ctc.CTCLoss(xr, targets, C - 1)
    loss.backward()
out["operator_loss_err"] = float(abs(loss.item() - want_loss.mean()) / abs(want_loss.mean()))
print("RESULT " + json.dumps(out))
"""


def _run(spin):
    env = dict(os.environ)
    if spin is not None:
        env["WFL_CTC_MITM_SPIN"] = str(spin)
    res = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads([l for l in res.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_ctc_launch_that_waits_out_its_bound_gives_up_cleanly_and_the_batch_is_recomputed():
    out = _run(1)
    for name in ("cfg1", "ragged"):
        r = out[name]
        assert r["gave_up"], (name, r)                      # reported ...
        assert r["repaired"] == r["B"], (name, r)           # ... every utterance recomputed in the log domain ...
        assert r["finite"] and r["loss_err"] < 1e-4 and r["dx_err"] < 1e-4, (name, r)  # ... and correct
    assert out["operator_loss_err"] < 1e-4


def test_ctc_launch_with_its_normal_bound_reports_no_give_up():
    out = _run(None)
    for name in ("cfg1", "ragged"):
        r = out[name]
        assert not r["gave_up"] and r["finite"] and r["loss_err"] < 1e-4 and r["dx_err"] < 1e-4, (name, r)
