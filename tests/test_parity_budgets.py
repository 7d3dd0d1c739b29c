"""CPU checks of the enumerated tolerance budgets the GPU parity tests add to their base gates (tests/helpers.py):
a budget must be ZERO where the ill-conditioning it accounts for is absent, and it must cover what fp32 does to the
reference's OWN math where it is present -- measured here as the difference between the oracle evaluated in fp32 (bit-equal
to the reference, tests/test_oracle_vs_reference.py) and the same oracle evaluated in fp64."""
import numpy as np
import pytest
import torch

from helpers import policy_saturation_budget, synth_batch
from oracle.dsact_oracle import DsactOracle, default_config, draw_noise


class _ZeroGrad:
    def __init__(self, ps):
        self.ps = ps

    def zero_grad(self):
        for p in self.ps:
            p.grad = None


def _fp64_twin(orc, cfg):
    o = DsactOracle(cfg, state_dict=orc.state_dict())
    for n in o.p:
        o.p[n] = [t.detach().double().requires_grad_(t.requires_grad) for t in o.p[n]]
    o.log_alpha = o.log_alpha.detach().double().requires_grad_(True)
    o.act_high, o.act_low = o.act_high.double(), o.act_low.double()
    o.opt = {"q1": _ZeroGrad(o.p["q1"]), "q2": _ZeroGrad(o.p["q2"]), "policy": _ZeroGrad(o.p["policy"]),
             "alpha": _ZeroGrad([o.log_alpha])}
    return o


@pytest.mark.parametrize("va,pa,eps_scale,saturated", [
    ("elu", "selu", 1.0, True),     # the case that missed the 3e-5 gate on the GPU (one action at 0.39992 of a 0.4 limit)
    ("gelu", "gelu", 1.0, True),
    ("gelu", "gelu", 0.3, False),   # same batch, smaller noise: no action near its limit
    ("elu", "selu", 1.3, True),
])
def test_policy_saturation_budget(va, pa, eps_scale, saturated):
    O, A, hid, B = 11, 3, (96, 40), 50
    torch.manual_seed(0)
    cfg = default_config(O, A, hid, act_limit=0.4, value_act=va, policy_act=pa)
    orc = DsactOracle(cfg)
    data = synth_batch(np.random.default_rng(5), B, O, A, lim=0.4, p_done=0.05)
    torch.manual_seed(1000)
    noise = draw_noise(B, A)
    noise = {k: (v * eps_scale if k == "eps_new" else v) for k, v in noise.items()}
    orc.compute_gradient(data, noise)
    g32 = torch.cat([t.grad.reshape(-1) for t in orc.p["policy"]]).numpy().astype(np.float64)
    budget = policy_saturation_budget(orc, data, noise, B)
    assert budget.shape == g32.shape and np.isfinite(budget).all() and (budget >= 0).all()
    o64 = _fp64_twin(orc, cfg)
    o64.compute_gradient({k: v.double() for k, v in data.items()},
                         {k: (v.double() if torch.is_tensor(v) else v) for k, v in noise.items()})
    g64 = torch.cat([t.grad.reshape(-1) for t in o64.p["policy"]]).numpy()
    err = np.abs(g32 - g64)
    base = 1e-9 + 3e-5 * np.abs(g64).max()      # the GPU tests' base gate on grad.policy
    if not saturated:
        assert budget.max() == 0.0              # nothing is added to the gate when no action is near its limit
        assert err.max() <= base
    else:
        assert budget.max() > 0.0
        # fp32 evaluation of the reference's own formulae stays inside base + budget with room to spare ...
        assert (err <= base + budget).all() and (err / (base + budget)).max() < 0.5
    if (va, pa, eps_scale) == ("elu", "selu", 1.0):
        # ... and NOT inside the base gate alone: the gate without the budget is tighter than fp32 itself on this batch
        assert (err > base).sum() > 0
