"""DSAC_V1 with the CNN approximators (example_train/dsacv1_cnn_carracing_offasync.py: dsac_v1.py:140-279 over
networks/cnn.py:151-240,383-461) on the HIP path -- four conv stacks instead of DSAC_V2's six, `k_loss_v1` over the twin-trunk
rows -- against oracle/dsac_v1_oracle_cnn.py, which tests/test_oracle_vs_reference.py pins bit-exact to the live reference.
Gates as in test_hip_cnn_parity.py: tb_info 1e-4 absolute, parameters / targets 1e-5 (no ReLU kink on the committed seeds:
asserted through the activations' distance from zero is not needed at batch 4 -- a kink would show as a 1e-3 miss)."""
import numpy as np
import pytest
import torch

from helpers import hip_kwargs
from oracle.dsac_v1_oracle import V1_TB_KEYS, draw_noise_v1
from oracle.dsac_v1_oracle_cnn import DsacV1CnnOracle
from oracle.dsact_oracle_cnn import cnn_config, synth_image_batch

pytestmark = pytest.mark.gpu


def make_pair(obs_shape, A, conv_type, B, seed=0, bound=True):
    from dsac_v1_hip import DSAC_V1_HIP

    kw = hip_kwargs(tuple(obs_shape), A, (256, 256, 256), B, act_limit=1.0, strict_rng=True, algorithm="DSAC_V1_HIP", TD_bound=10,
                    bound=bound)
    for key in ("value", "policy"):
        kw[key + "_func_type"], kw[key + "_conv_type"] = "CNN", conv_type
        kw.pop(key + "_hidden_sizes")
    torch.manual_seed(seed)
    alg = DSAC_V1_HIP(**kw)
    cfg = cnn_config(obs_shape, A, conv_type, TD_bound=10, bound=bound)
    orc = DsacV1CnnOracle(cfg, state_dict={k: v.cpu() for k, v in alg.networks.state_dict().items()})
    return alg, orc, cfg


@pytest.mark.parametrize("conv_type,obs_shape,B,bound", [("type_2", (3, 96, 96), 4, True), ("type_2", (3, 96, 96), 8, False),
                                                         ("type_1", (4, 84, 84), 4, True), ("type_2", (3, 96, 96), 16, True),
                                                         ("type_2", (3, 96, 96), 32, False)])
def test_v1_cnn_against_oracle(conv_type, obs_shape, B, bound):
    A = 3
    alg, orc, cfg = make_pair(obs_shape, A, conv_type, B, bound=bound)
    sd0 = alg.networks.state_dict()
    assert list(sd0.keys())[:2] == ["log_alpha", "q.conv.0.weight"] and "q_target.mean.0.weight" in sd0
    # batch % 16 == 0 with equal trunk widths (type_2): the twin trunks run as row-slice chain units, else on the stage tiles
    assert alg.engine.layout.n_critics == 1 and alg.engine.chain_active == (B % 16 == 0 and conv_type == "type_2")
    for it in range(3):
        data = synth_image_batch(cfg, B, seed=it)
        torch.manual_seed(500 + it)
        noise = draw_noise_v1(B, A)
        torch.manual_seed(500 + it)
        tb = alg.local_update(data, it)
        ref = orc.local_update(data, noise, it)
        assert list(tb.keys()) == V1_TB_KEYS
        for k in V1_TB_KEYS[:-1]:
            assert abs(float(tb[k]) - float(ref[k])) <= 1e-4, (it, k, float(tb[k]), float(ref[k]))
        sd, osd = alg.networks.state_dict(), orc.state_dict()
        assert list(sd.keys()) == list(osd.keys())
        worst = max((float((sd[k].cpu() - osd[k]).abs().max()), k) for k in sd)
        assert worst[0] <= 1e-5, (it, worst)
    assert np.isfinite(alg.engine.online.cpu().numpy()).all()
    # the acting forward on an image and the evaluation surface work on the attached container
    obs = synth_image_batch(cfg, 2, seed=9)["obs"]
    lg = alg.networks.policy(obs)
    assert tuple(lg.shape) == (2, 2 * A) and torch.isfinite(lg).all()
